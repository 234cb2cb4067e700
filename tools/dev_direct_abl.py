"""Dev helper: compile-time ablations of the direct conv kernel on the 32->32 bf16 level-1 conv of config 2
(results of the ablated variants are wrong on purpose; only their time is of interest)."""
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
plain = lv.nbr.clone()                                    # the same table without the attached column form
if len(sys.argv) > 1 and sys.argv[1] == "ct":             # column-form kernel: requests one phase ahead (default) / words only / neither
    ref = None
    for rnd in range(2):
        for mode, nm in ((0, "column form, rulebook words one tile ahead"), (13, "column form, nothing ahead")):
            hook(mode)
            for _ in range(5): run()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            if ref is None: ref = out.clone()
            print(f"{nm:50s} {e0.elapsed_time(e1) / 20:.3f} ms  identical={bool(torch.equal(out, ref))}")
    hook(0)
    sys.exit(0)
run = lambda: ops.conv_fwd(x, w, plain, lv.n, out=out, residual=res)
names = {0: "full kernel (16 waves, G 3)", 1: "no gathers", 2: "no MFMA", 3: "no output stores", 4: "no rulebook loads (identity rows)",
         5: "no rulebook loads, no gathers", 6: "no gathers, no stores", 7: "8 waves per workgroup", 8: "G = 9 taps per group", 9: "G = 1", 100: "full kernel again"}
for _ in range(30): run()
for mode, nm in names.items():
    hook(mode % 100)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{nm:40s} {e0.elapsed_time(e1) / 20:.3f} ms")
hook(0)

# segment timers (mode 10): where a wave's cycles go per 32-row tile
tmf = lib.tl_dev_direct_tm; tmf.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 8)()
hook(10); run(); torch.cuda.synchronize(); tmf(buf)
for _ in range(5): run()
torch.cuda.synchronize(); tmf(buf); hook(0)
v = [int(x) for x in buf]; tiles = max(v[5], 1)
names_tm = ["rulebook entries (+ older stores)", "issuing gathers", "waiting for gathers", "LDS fragments + MFMA", "epilogue"]
tot = sum(v[:5])
print(f"per tile and wave: {tot / tiles:.0f} cycles")
for nm, c in zip(names_tm, v[:5]):
    print(f"  {nm:36s} {c / tiles:8.0f} cycles  {100.0 * c / tot:5.1f} %")
