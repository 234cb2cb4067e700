"""Dev (developer build: python -m treelearn_amd.build --dev): what could a present-pairs-only contraction gain at level 2?  The blk_floor
method (tools/dev_blk_floor.py, DESIGN R5.4) on the product's level-2 kernel, k_conv_streamq<27, 2, 1, ..> (64 -> 64 on the config-2 level-2
rulebook, 16.3 of 27 (row, tap) pairs present per row):
  product     all 27 taps of every 32-row tile gathered and contracted
  (a)         all 27 gathers, MFMAs for 16 taps            = an ideal present-pairs kernel that still addresses every tap
  (b)         gathers AND MFMAs for 16 taps                = the floor of ANY formulation that skips absent pairs, bookkeeping at zero cost
  (b')        as (b) with every one of the 16 x n entries present (absent entries point at the row itself): a compacted list has no holes,
              so its gathers all move bytes -- the honest form of (b)
  (c)         (b) / (b') with ONE output view
each with one view (plain store), with the residual + second (activated) view the ResidualBlock's second conv writes, and with the view alone.
(Results of the ablated runs are wrong on purpose.)  Stop rule (round-5 verdict, item 2): build a per-tap-compacted kernel only if (b) <= 0.24 ms."""
import ctypes, os, sys
os.environ["TL_NO_COMPACT"] = "1"          # the 27-entry table form (the K = 16 instantiation reads its first 16 rows)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import _hip, ops
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_tile

lib = _hip.lib()
if lib.tl_set_tuning(b"win", 0) != 0:
    sys.exit("needs the developer build (python -m treelearn_amd.build --dev)")
hook = lib.tl_dev_streamq_tm
hook.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
cfg = CONFIGS["config2"]
t = make_tile(**cfg, seed=0)
pts = torch.from_numpy(t["points"]).cuda(); bid = torch.zeros(len(pts), dtype=torch.int64, device="cuda")
g = build_geometry(pts, bid, 1, cfg["voxel"], 7, [500, 500, 1000])
lv = g.levels[1]; C = 64; n = lv.n
tab = lv.nbr.clone()                                         # i32[27][n], -1 = absent
present = (tab >= 0)
print(f"config-2 level 2: {n} rows, present (row, tap) pairs per row {present.sum().item() / n:.2f} of 27; of the first 16 taps {present[:16].sum().item() / n:.2f}")
full = torch.where(present, tab, torch.arange(n, device="cuda", dtype=torch.int32)[None, :]).contiguous()
torch.manual_seed(0)
x = torch.randn(n, C, device="cuda").to(torch.bfloat16)
w = ops.pack_weight(torch.randn(C, 3, 3, 3, C, device="cuda") * 0.05, torch.bfloat16)
res = torch.randn(n, C, device="cuda").to(torch.bfloat16); out = torch.empty_like(x); out2 = torch.empty_like(x)
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.1


def timeit(f, reps=30, warm=8):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


forms = (("one view", lambda tb: ops.conv_fwd(x, w, tb, n, out=out)),
         ("residual, one view", lambda tb: ops.conv_fwd(x, w, tb, n, out=out, residual=res)),
         ("residual + activated second view", lambda tb: ops.conv_fwd(x, w, tb, n, out=out, residual=res, out2=(out2, sc, sh, True))))
variants = ((0, tab, "product kernel (27 taps)"), (21, tab, "(a) 27 gathers, MFMAs for 16 taps"), (20, tab, "(b) gathers + MFMAs for 16 taps"),
            (20, full, "(b') as (b), all 16 x n entries present"), (0, full, "product kernel, all 27 x n entries present"))
for rep in range(2):
    for mode, tb, what in variants:
        hook(mode, None)
        ts = [timeit(lambda f=f: f(tb)) for _, f in forms]
        print(f"{what:44s} " + "   ".join(f"{nm}: {v:.4f} ms" for (nm, _), v in zip(forms, ts)), flush=True)
hook(0, None)
