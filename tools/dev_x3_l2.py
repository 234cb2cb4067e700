"""Dev: the parity-fast mode's level-2 / level-3 convs (fp32 rows, split-bf16 contraction) on the config-2 rulebooks: the fragment-shape gather
kernel (tl_conv_stream.hip, X3) against the quad-coalesced form (tl_conv_streamq.hip, X3) and its variants (tl_set_tuning "streamq_x3":
0 off, 1 = the shipped choice, 2 / 3 = prefetch depth 2 / 1 with the two-step loop for every shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import _hip, geometry as G, ops
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

L = _hip.lib()
b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
g = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 1, 0.1, 7, [500, 500, 1000])
torch.manual_seed(0)


def timeit(f, reps=20, warm=4):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for li, cin, cout in ((1, 64, 64), (1, 128, 64), (2, 96, 96), (2, 192, 96)):
    lv = g.levels[li]; n = lv.n
    x = torch.randn(n, cin, device="cuda"); res = torch.randn(n, cout, device="cuda")
    ops.PACK_X3 = True
    w = ops.pack_weight(torch.randn(cout, 3, 3, 3, cin, device="cuda") * 0.05, torch.float32)
    ops.PACK_X3 = False
    o1 = torch.empty(n, cout, device="cuda"); o2 = torch.empty_like(o1)
    sc = torch.rand(cout, device="cuda") + 0.5; sh = torch.randn(cout, device="cuda") * 0.1
    for mode in (0, 1, 2, 3, 0, 1):
        _hip.check(L.tl_set_tuning(b"streamq_x3", mode), "streamq_x3")
        t0 = timeit(lambda: ops.conv_fwd(x, w, lv.nbr, n, out=o1))
        t1 = timeit(lambda: ops.conv_fwd(x, w, lv.nbr, n, out=o1, residual=res, out2=(o2, sc, sh, True)))
        print(f"level {li + 1} {cin:3d} -> {cout:3d} ({n} rows)  streamq_x3 = {mode}:  one view {t0:.4f} ms   residual + activated second view {t1:.4f} ms", flush=True)
_hip.check(L.tl_set_tuning(b"streamq_x3", 1), "streamq_x3")
