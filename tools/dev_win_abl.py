"""Ablations of the window conv kernel on the real level-2 rulebook (dev build: python -m treelearn_amd.build --dev).
    python tools/dev_win_abl.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import _hip, ops
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_batch, make_tile
b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
_hip.WIN_KERNEL = True
g = build_geometry(b["coords"].cuda(), b["batch_ids"].cuda(), 1, 0.1, 7, [500, 500, 1000])
L = _hip.lib(); hook = L.tl_dev_win_abl; hook.argtypes = [ctypes.c_int]
_hip.check(L.tl_set_tuning(b"win", 2), "win"); _hip.check(L.tl_set_tuning(b"win_min_rows", 0), "wmr")
lv = g.levels[1]
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
x = torch.randn((lv.n, 64), device="cuda", generator=gen).bfloat16()
res = torch.randn((lv.n, 64), device="cuda", generator=gen).bfloat16()
w = ops.pack_weight(torch.randn((64, 3, 3, 3, 64), device="cuda", generator=gen) / (27 * 64) ** 0.5, torch.bfloat16)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
names = {0: "full kernel", 1: "no window DMA", 16: "no weight DMA", 17: "no DMA at all", 2: "no MFMA", 4: "no LDS fragment reads", 6: "no MFMA, no fragment reads",
         8: "no output stores", 32: "no index loads", 64: "no step barrier", 23: "no DMA, no MFMA, no fragment reads", 55: "only loop + index math + epilogue",
         128: "no decode / address arithmetic / minima", 136: "... and no output stores"}
for mode, name in names.items():
    hook(mode)
    for variant, fn in (("residual + 1 view", lambda: ops.conv_fwd(x, w, lv.nbr, lv.n, residual=res)), ("plain", lambda: ops.conv_fwd(x, w, lv.nbr, lv.n))):
        print(f"ABL {mode:3d} {name:40s} {variant:18s} {timeit(fn):.3f} ms", flush=True)
hook(0)
