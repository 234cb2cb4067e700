"""Dev helper: time ONE conv shape on the real config-2 rulebooks (for kernel tuning / rocprofv3 --pmc)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from treelearn_amd import ops, _hip
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_tile

for key in ("bf16_depth", "bf16_units", "small_rows", "small_mode", "dbg", "direct", "stream", "streamq", "stream_rb"):
    if os.environ.get("TL_" + key.upper()):
        _hip.check(_hip.lib().tl_set_tuning(key.encode(), int(os.environ["TL_" + key.upper()])), key)
level = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # 0-based level
dtype = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
pro = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
cfg = CONFIGS["config2"]
t = make_tile(**cfg, seed=0)
pts = torch.from_numpy(t["points"]).cuda(); bid = torch.zeros(len(pts), dtype=torch.int64, device="cuda")
g = build_geometry(pts, bid, 1, cfg["voxel"], 7, [500, 500, 1000])
lv = g.levels[level]; C = 32 * (level + 1)
x = torch.randn(lv.n, C, device="cuda").to(dtype)
w = ops.pack_weight(torch.randn(C, 3, 3, 3, C, device="cuda") * 0.05, dtype)
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.1
res = torch.randn(lv.n, C, device="cuda").to(dtype)
out = torch.empty_like(x)
def run():
    if pro: ops.conv_fwd(x, w, lv.nbr, lv.n, out=out, in_scale=sc, in_shift=sh, in_relu=True, out_scale=sc, out_shift=sh, out_relu=True)
    else: ops.conv_fwd(x, w, lv.nbr, lv.n, out=out, residual=res)
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
pairs = int((lv.nbr >= 0).sum())
print(f"level {level+1} C={C} n={lv.n} pairs={pairs} dtype={dtype} pro={pro}: {ms:.3f} ms  {2.0*pairs*C*C/ms/1e9:.1f} TFLOP/s  gather {pairs*C*x.element_size()/ms/1e6:.0f} GB/s")
