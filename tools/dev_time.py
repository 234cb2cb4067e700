"""Dev helper: stage-by-stage timing of one forward on the GPU."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from treelearn_amd.synth import random_state_dict
from treelearn_amd import ops
from treelearn_amd.model import TreeLearn
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

from treelearn_amd import _hip
for key in ("bf16_depth", "bf16_units", "small_rows", "small_mode", "dbg", "direct", "stream", "streamq", "stream_rb"):
    if os.environ.get("TL_" + key.upper()):
        _hip.check(_hip.lib().tl_set_tuning(key.encode(), int(os.environ["TL_" + key.upper()])), key)
        print("tuning", key, os.environ["TL_" + key.upper()])
name = sys.argv[1] if len(sys.argv) > 1 else "config2"
dtype = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
cfg = CONFIGS[name]
SS = [500, 500, 1000] if cfg["voxel"] >= 0.1 else None
t0 = time.time(); tile = make_tile(**cfg, seed=0); batch = make_batch([tile]); print("tile", time.time() - t0, flush=True)
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=SS, voxel_size=cfg["voxel"], compute_dtype=dtype)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
torch.cuda.synchronize(); print("model ready", flush=True)
for it in range(3):
    t0 = time.time()
    geom = build_geometry(g["coords"], g["batch_ids"], 1, cfg["voxel"], 7, SS)
    torch.cuda.synchronize(); print("geometry", time.time() - t0, [l.n for l in geom.levels], flush=True)
ops.PROFILE = []
with torch.no_grad():
    t0 = time.time(); out = model(g, return_loss=False); torch.cuda.synchronize(); print("forward#1", time.time() - t0, flush=True)
    ops.PROFILE = []
    t0 = time.time(); out = model(g, return_loss=False); torch.cuda.synchronize(); print("forward#2 (profiled)", time.time() - t0, flush=True)
recs = ops.PROFILE; ops.PROFILE = None
rows = {}
for e0, e1, m in recs:
    key = (m["K"], m["Cin"], m["Cout"], m["n_out"])
    ms = e0.elapsed_time(e1)
    pairs = int((m["table"] >= 0).sum()) if m["table"] is not None else m["n_out"]
    r = rows.setdefault(key, [0, 0.0, 0.0]); r[0] += 1; r[1] += ms; r[2] += 2.0 * pairs * m["Cin"] * m["Cout"]
tot = 0
for key, (cnt, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"K={key[0]:2d} Cin={key[1]:3d} Cout={key[2]:3d} n_out={key[3]:8d} x{cnt:2d}: {ms:9.3f} ms  {fl / ms / 1e9:8.2f} TFLOP/s")
    tot += ms
print("conv total ms", tot)
with torch.no_grad():
    for _ in range(2):
        t0 = time.time(); out = model(g, return_loss=False); torch.cuda.synchronize(); print("forward", time.time() - t0, flush=True)
