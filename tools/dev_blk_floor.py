"""Dev (developer build: python -m treelearn_amd.build --dev): what could a present-pairs-only contraction gain at level 1?  Times the
product's staged-unit kernel (k_conv_blk, 32 -> 32 on the config-2 level-1 rulebook) with its tap loop ablated (tl_set_tuning "dbg"):
  0   the product kernel: all 27 taps of every 32-row tile contracted (5.5 of 27 (row, tap) pairs exist on average)
  16  MFMAs for 5 taps only, all LDS reads kept            = an ideal present-pairs kernel that still reads every tap's operands
  48  MFMAs AND LDS operand reads for 5 taps only          = the floor of ANY formulation that skips absent pairs, bookkeeping at zero cost
  4   no MFMAs at all                                      = everything but the matrix work (staging DMA, rulebook, epilogue, stores)
(results of the ablated runs are wrong on purpose).  Verdict item (round 4, #3): land a present-pairs kernel only if <= 0.085 ms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import _hip, geometry as G, ops
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

L = _hip.lib()
if L.tl_set_tuning(b"win", 0) != 0:
    sys.exit("needs the developer build (python -m treelearn_amd.build --dev)")
b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
c, bi = b["coords"].cuda().float(), b["batch_ids"].cuda().long()
blk = G.build_geometry(c, bi, 1, 0.1, 7, [500, 500, 1000], blocked=True)
r = blk.levels[0].nbr; n = blk.levels[0].n
torch.manual_seed(0)
x = torch.randn(n, 32, device="cuda").bfloat16(); res = torch.randn(n, 32, device="cuda").bfloat16()
w = ops.pack_weight(torch.randn(32, 3, 3, 3, 32, device="cuda") * 0.1, torch.bfloat16)
o1 = torch.empty_like(x)


def timeit(f, reps=30, warm=5):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"config-2 level 1: {n} rows, present (row, tap) pairs per row {r.count_pairs() / n:.2f} of 27")
for rep in range(2):
    for dbg, what in ((0, "product kernel"), (16, "MFMAs for 5 taps, all LDS reads"), (48, "MFMAs and LDS reads for 5 taps"), (4, "no MFMAs")):
        _hip.check(L.tl_set_tuning(b"dbg", dbg), "dbg")
        t0 = timeit(lambda: ops.conv_fwd(x, w, r, n, out=o1))
        t1 = timeit(lambda: ops.conv_fwd(x, w, r, n, out=o1, residual=res))
        print(f"dbg {dbg:2d}  {what:36s} plain {t0:.4f} ms   with residual {t1:.4f} ms", flush=True)
_hip.check(L.tl_set_tuning(b"dbg", 0), "dbg")
