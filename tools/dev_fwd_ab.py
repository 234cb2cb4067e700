"""Dev: same-box A/B of the whole bf16 forward (config 2) under direct-kernel developer modes:
0 = default, 14 = 64->32 conv on the 27-entry table, 15 = input conv on the 27-entry table, 13 = 32->32 without look-ahead."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd.synth import random_state_dict
from treelearn_amd import _hip
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile
hook = _hip.lib().tl_dev_direct_abl; hook.argtypes = [ctypes.c_int]
cfg = CONFIGS["config2"]
b = make_batch([make_tile(**cfg, seed=0)])
g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=cfg["voxel"], compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
modes = [int(a) for a in sys.argv[1:]] or [0, 14, 15, 13]
with torch.no_grad():
    for _ in range(5): model(g, return_loss=False)
    for rnd in range(3):
        for m in modes:
            hook(m)
            for _ in range(3): model(g, return_loss=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): model(g, return_loss=False)
            torch.cuda.synchronize()
            print(f"mode {m:3d}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms", flush=True)
hook(0)
