"""Dev helper: compile-time ablations of the stream-q conv kernel on the 64->64 bf16 level-2 conv of config 2
(results of the ablated variants are wrong on purpose; only their time is of interest)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import ops, _hip
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_tile

lib = _hip.lib()
hook = lib.tl_dev_streamq_tm
hook.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
cfg = CONFIGS["config2"]
t = make_tile(**cfg, seed=0)
pts = torch.from_numpy(t["points"]).cuda(); bid = torch.zeros(len(pts), dtype=torch.int64, device="cuda")
g = build_geometry(pts, bid, 1, cfg["voxel"], 7, [500, 500, 1000])
lv = g.levels[1]; C = 64
x = torch.randn(lv.n, C, device="cuda").to(torch.bfloat16)
w = ops.pack_weight(torch.randn(C, 3, 3, 3, C, device="cuda") * 0.05, torch.bfloat16)
res = torch.randn(lv.n, C, device="cuda").to(torch.bfloat16); out = torch.empty_like(x)
run = lambda: ops.conv_fwd(x, w, lv.nbr, lv.n, out=out, residual=res)
names = {0: "full kernel (8 waves, RB 1, depth 2)", 2: "no transposition", 3: "no gathers", 4: "no barrier", 6: "no gathers, no transposition",
         8: "prefetch depth 3", 9: "prefetch depth 1", 10: "4 waves, RB 1", 11: "4 waves, RB 2", 12: "8 waves, RB 2",
         13: "4 waves, RB 2, depth 1", 14: "4 waves, RB 2, depth 3", 15: "4 waves, RB 2, no gathers", 100: "full kernel again"}
for _ in range(30): run()
for mode, nm in names.items():
    hook(mode % 100, None)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{nm:32s} {e0.elapsed_time(e1) / 20:.3f} ms")
hook(0, None)
