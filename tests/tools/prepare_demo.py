"""Measure plot preparation on a synthetic raw cloud: device voxel down-sample + verticality, beside the
numpy/scipy restatement (oracle/prepare.py, one core) on a bounded sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from treelearn_amd.synth import CONFIGS, make_tile
from treelearn_amd.util.prepare import compute_features, voxelize

t = make_tile(**CONFIGS["config2"], seed=0)
rng = np.random.default_rng(0)
raw = np.vstack([t["points"].astype(np.float64) + rng.normal(0, 0.02, size=t["points"].shape) for _ in range(3)])     # ~3 raw points per voxel
print(f"raw cloud {len(raw)} points")
d = torch.from_numpy(raw).cuda()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    down, trace = voxelize(d, 0.1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    feats = compute_features(down[:, :3], 0.6)
    torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"voxelize: {len(down)} voxels, {(t1 - t0) * 1e3:.1f} ms = {len(raw) / (t1 - t0) / 1e6:.0f} Mpoints/s;  "
      f"verticality: {(t2 - t1) * 1e3:.1f} ms = {len(down) / (t2 - t1) / 1e6:.1f} Mpoints/s")
from oracle import prepare as op
m = 60000
t0 = time.perf_counter(); op.voxelize(raw[:m], 0.1); t1 = time.perf_counter()
sub = down[:, :3].cpu().numpy(); sub = sub[(np.abs(sub[:, 0]) < 4) & (np.abs(sub[:, 1]) < 4)]
t2 = time.perf_counter(); op.verticality(sub, 0.6); t3 = time.perf_counter()
print(f"CPU restatement (1 core): voxelize {m / (t1 - t0) / 1e6:.3f} Mpoints/s on {m} points; verticality {len(sub) / (t3 - t2) / 1e6:.4f} Mpoints/s on {len(sub)} points")
