"""End-to-end demo of the hot path on synthetic tiles: tile loop (H2D + forward + device-side inner filter + D2H)
-> grouping (DBSCAN / HDBSCAN on the GPU).  Mirrors tools/pipeline/pipeline.py:66-94 of the reference minus file I/O.

    python tools/pipeline_demo.py [n_tiles] [extent_m] [bf16|fp32]
"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from treelearn_amd.synth import random_state_dict
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import make_batch, make_tile
from treelearn_amd.util import get_pointwise_preds, get_instances

from bench import host_cores
torch.set_num_threads(host_cores())          # the box exposes 256 CPUs but a 16-CPU quota: an oversubscribed OpenMP pool stalls CPU ops
n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
extent = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
dtype = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "fp32") else torch.bfloat16
tiles = []
for s in range(n_tiles):
    t = make_tile(extent=extent, voxel=0.1, n_trees=max(1, int(64 * (extent / 40) ** 2)), fill=0.10, seed=s)
    t["center"] = np.array([8.0 * (s % 8), 8.0 * (s // 8), 0.0])
    b = make_batch([t], inner_square_edge_length=8.0)
    tiles.append({k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()})
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=dtype)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
get_pointwise_preds(model, tiles[:1], dict(voxel_size=0.1))             # warm-up
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    res = get_pointwise_preds(model, tiles, dict(voxel_size=0.1))
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"pass {rep}: {dt*1e3/n_tiles:.1f} ms/tile, reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
npts = sum(t["coords"].shape[0] for t in tiles)
print(f"tile loop: {n_tiles} tiles, {npts} points, {dt*1e3/n_tiles:.1f} ms/tile incl. H2D/D2H = {npts/dt/1e6:.1f} Mpoints/s; inner points kept {res[0].shape[0]}")
sem, seml, off, offl, coords, inst, bb, feats = res
cfg = dict(tree_conf_thresh=0.5, tau_vert=0.6, tau_off=4, tau_group=0.15, tau_min=50, use_hdbscan=False)
# random-init weights give meaningless offsets: group on ground-truth-like offsets so the demo clusters trunks
t0 = time.time(); p1 = get_instances(coords, offl, np.stack([-seml * 4.0 + 2, seml * 4.0 - 2], 1).astype(np.float32), cfg, feats[:, -1], 0, 0, -1, 1); t1 = time.time() - t0
cfg["use_hdbscan"] = True
t0 = time.time(); p2 = get_instances(coords, offl, np.stack([-seml * 4.0 + 2, seml * 4.0 - 2], 1).astype(np.float32), cfg, feats[:, -1], 0, 0, -1, 1); t2 = time.time() - t0
print(f"grouping: dbscan {t1*1e3:.1f} ms -> {p1.max()} trees; hdbscan {t2*1e3:.1f} ms -> {p2.max()} trees (points {len(coords)})")
