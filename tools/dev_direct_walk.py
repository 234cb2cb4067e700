"""Dev: tile walk of the direct conv kernel on the level-1 32->32 conv of config 2 -- default (workgroup b takes tiles
16 b .. 16 b + 15, then strides by the grid) vs XCD-sliced (every XCD walks its own contiguous eighth).  With
`rocprofv3 --pmc FETCH_SIZE` the per-dispatch traffic of the two can be compared (mode switches after 10 launches each)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import ops, _hip
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_tile

lib = _hip.lib()
hook = lib.tl_dev_direct_abl; hook.argtypes = [ctypes.c_int]
cfg = CONFIGS["config2"]
t = make_tile(**cfg, seed=0)
pts = torch.from_numpy(t["points"]).cuda(); bid = torch.zeros(len(pts), dtype=torch.int64, device="cuda")
g = build_geometry(pts, bid, 1, cfg["voxel"], 7, [500, 500, 1000])
lv = g.levels[0]; C = 32
x = torch.randn(lv.n, C, device="cuda").to(torch.bfloat16)
w = ops.pack_weight(torch.randn(C, 3, 3, 3, C, device="cuda") * 0.05, torch.bfloat16)
res = torch.randn(lv.n, C, device="cuda").to(torch.bfloat16); out = torch.empty_like(x)
run = lambda: ops.conv_fwd(x, w, lv.nbr, lv.n, out=out, residual=res)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ref = None
for rnd in range(2):
    for walk in (0, 1):
        hook(1000 + walk)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        print(f"walk {walk}: {e0.elapsed_time(e1) / reps:.3f} ms  identical={bool(torch.equal(out, ref))}")
hook(1000)
