"""Dev: the 64 -> 32 level-1 conv (2C -> C of the decoder) on the direct kernel: 27-entry table vs column form with the
rulebook words requested one tile ahead (default; ablation mode 14 = the table form)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import ops, _hip
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_tile
lib = _hip.lib(); hook = lib.tl_dev_direct_abl; hook.argtypes = [ctypes.c_int]
cfg = CONFIGS["config2"]; t = make_tile(**cfg, seed=0)
pts = torch.from_numpy(t["points"]).cuda(); bid = torch.zeros(len(pts), dtype=torch.int64, device="cuda")
g = build_geometry(pts, bid, 1, cfg["voxel"], 7, [500, 500, 1000]); lv = g.levels[0]
x = torch.randn(lv.n, 64, device="cuda").to(torch.bfloat16)
w = ops.pack_weight(torch.randn(32, 3, 3, 3, 64, device="cuda") * 0.05, torch.bfloat16)
sc = torch.rand(32, device="cuda") + 0.5; sh = torch.randn(32, device="cuda") * 0.1
out = torch.empty(lv.n, 32, device="cuda", dtype=torch.bfloat16)
run = lambda: ops.conv_fwd(x, w, lv.nbr, lv.n, out=out, out_scale=sc, out_shift=sh, out_relu=True)
ref = None
for rnd in range(2):
    for mode, nm in ((14, "27-entry table"), (0, "column form, words one tile ahead")):
        hook(mode)
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        print(f"{nm:40s} {e0.elapsed_time(e1) / 20:.3f} ms  identical={bool(torch.equal(out, ref))}")
hook(0)

# the 4 -> 32 input conv: 27-entry table (mode 15) vs column form with look-ahead (default)
x4 = torch.randn(lv.n, 4, device="cuda").to(torch.bfloat16)
w4 = ops.pack_weight(torch.randn(32, 3, 3, 3, 4, device="cuda") * 0.2, torch.bfloat16)
o4 = torch.empty(lv.n, 32, device="cuda", dtype=torch.bfloat16); o4b = torch.empty_like(o4)
run4 = lambda: ops.conv_fwd(x4, w4, lv.nbr, lv.n, out=o4, out2=(o4b, sc, sh, True))
ref = None
for rnd in range(2):
    for mode, nm in ((15, "in4, 27-entry table"), (0, "in4, column form, words one tile ahead")):
        hook(mode)
        for _ in range(5): run4()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run4()
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref = o4.clone()
        print(f"{nm:40s} {e0.elapsed_time(e1) / 20:.3f} ms  identical={bool(torch.equal(o4, ref))}")
hook(0)
