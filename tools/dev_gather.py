"""Dev: raw gather-bandwidth ceiling of the rulebook access pattern (no LDS, no MFMA)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import _hip
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_tile
L = _hip.lib()
fns = {}
for nm in ("tl_dev_gather_bench", "tl_dev_gather_frag", "tl_dev_gather_quad"):
    f = getattr(L, nm); f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    fns[nm] = f
cfg = CONFIGS["config2"]
t = make_tile(**cfg, seed=0)
pts = torch.from_numpy(t["points"]).cuda(); bid = torch.zeros(len(pts), dtype=torch.int64, device="cuda")
g = build_geometry(pts, bid, 1, cfg["voxel"], 7, [500, 500, 1000])
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
for level in (0, 1, 2):
    lv = g.levels[level]; C = 32 * (level + 1)
    x = torch.randn(lv.n, C, device="cuda").to(torch.bfloat16)
    pairs = int((lv.nbr >= 0).sum())
    for nm, tif in [(a, b) for a in fns for b in (1, 3, 9)]:
        fn = fns[nm]
        if C == 96 and tif == 9: continue
        if nm.endswith("quad") and C != 64: continue
        for _ in range(2): fn(x.data_ptr(), C * 2, C * 2, lv.nbr.data_ptr(), 27, lv.n, tif, sink.data_ptr(), _hip.stream())
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn(x.data_ptr(), C * 2, C * 2, lv.nbr.data_ptr(), 27, lv.n, tif, sink.data_ptr(), _hip.stream())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{nm[7:]:12s} level {level+1} C={C} taps_in_flight={tif}: {ms:.3f} ms  valid-row gather {pairs*C*2/ms/1e6:.0f} GB/s  slot rate {27*lv.n*C*2/ms/1e6:.0f} GB/s (incl. absent)")
