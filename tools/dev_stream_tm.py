"""Dev helper: where do the waves of the stream conv kernels (64->64, bf16, level 2 of config 2) spend their cycles?
Per-step segments (summed over waves by the kernels' timing builds): 0 wait for A/weights + ds_write, 1 (transposition +)
LDS reads + MFMA issue, 2 gather issue, 3 barrier.  Usage: dev_stream_tm.py [q]   (q = the quad-gather form)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import ops, _hip
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_tile

lib = _hip.lib()
quad = len(sys.argv) > 1 and sys.argv[1] == "q"
hook = lib.tl_dev_streamq_tm if quad else lib.tl_dev_stream_tm
hook.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
_hip.check(lib.tl_set_tuning(b"streamq", 1 if quad else 0), "streamq")
cfg = CONFIGS["config2"]
t = make_tile(**cfg, seed=0)
pts = torch.from_numpy(t["points"]).cuda(); bid = torch.zeros(len(pts), dtype=torch.int64, device="cuda")
g = build_geometry(pts, bid, 1, cfg["voxel"], 7, [500, 500, 1000])
lv = g.levels[1]; C = 64
x = torch.randn(lv.n, C, device="cuda").to(torch.bfloat16)
w = ops.pack_weight(torch.randn(C, 3, 3, 3, C, device="cuda") * 0.05, torch.bfloat16)
res = torch.randn(lv.n, C, device="cuda").to(torch.bfloat16); out = torch.empty_like(x)
run = lambda: ops.conv_fwd(x, w, lv.nbr, lv.n, out=out, residual=res)
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"{'stream-q' if quad else 'stream'} 64->64 level 2: {e0.elapsed_time(e1) / 10:.3f} ms")
buf = (ctypes.c_ulonglong * 8)()
hook(1, buf)
run(); torch.cuda.synchronize()
hook(0, buf)
waves = buf[4]
names = ["wait A/W + ds_write", "LDS reads + MFMA", "gather issue", "barrier"]
tot = sum(buf[i] for i in range(4))
for i in range(4):
    print(f"  {names[i]:22s} {buf[i] / waves / 27:9.1f} clk/step/wave  {100.0 * buf[i] / tot:5.1f} %")
print(f"  total {tot / waves / 27:.1f} clk/step/wave over {waves} waves")
