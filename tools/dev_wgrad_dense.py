"""Dev: the dense-over-taps weight-gradient kernel (csrc/tl_wgrad_dense.hip) against the pair-list kernels of tl_wgrad.hip on the real
config-3 rulebooks (2 x 40 m crops): time of both and their difference.      python tools/dev_wgrad_dense.py [l1 l2 l3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import geometry as G, ops, _hip
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

tiles = [make_tile(**CONFIGS["config2"], seed=s) for s in (0, 1)]
b = make_batch(tiles)
geom = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 2, 0.1, 7, [500, 500, 1000])
L = _hip.lib()
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
only = set(a for a in sys.argv[1:] if a.startswith("l"))
for li, lv in enumerate(geom.levels[:4]):
    if only and f"l{li+1}" not in only: continue
    C = 32 * (li + 1)
    for ci, co in ((C, C), (2 * C, C)):
        x = torch.randn(lv.n, ci, device="cuda").bfloat16(); g = torch.randn(lv.n, co, device="cuda").bfloat16()
        L.tl_set_tuning(b"wgrad_dense", 0)
        ref = ops.conv_wgrad(x, g, lv.nbr, lv.n, 27); t0 = timeit(lambda: ops.conv_wgrad(x, g, lv.nbr, lv.n, 27))
        L.tl_set_tuning(b"wgrad_dense", 1); L.tl_set_tuning(b"wgrad_dma", 0)
        new = ops.conv_wgrad(x, g, lv.nbr, lv.n, 27); new2 = ops.conv_wgrad(x, g, lv.nbr, lv.n, 27)
        t1 = timeit(lambda: ops.conv_wgrad(x, g, lv.nbr, lv.n, 27))
        for var, gxo in ((1, 0), (2, 0)):
            L.tl_set_tuning(b"wgrad_dma", var); L.tl_set_tuning(b"wgrad_dense_gx", gxo)
            dma = ops.conv_wgrad(x, g, lv.nbr, lv.n, 27)
            t2 = timeit(lambda: ops.conv_wgrad(x, g, lv.nbr, lv.n, 27))
            print(f"    LDS-DMA form variant {var} slots {gxo or 'table'}: {t2:7.3f} ms, max |dma - dense| / max |dense| = {float((dma - new).abs().max() / new.abs().max()):.2e}")
        L.tl_set_tuning(b"wgrad_dma", 1); L.tl_set_tuning(b"wgrad_dense_gx", 0)
        err = float((new - ref).abs().max() / ref.abs().max())
        pairs = int((lv.nbr >= 0).sum())
        print(f"l{li+1} {ci:3d}->{co:3d} rows {lv.n:8d} pairs/row {pairs / lv.n:5.2f}: pair-list {t0:7.3f} ms, dense {t1:7.3f} ms "
              f"({2.0 * pairs * ci * co / t1 / 1e9:5.0f} TFLOP/s present, {2.0 * 27 * lv.n * ci * co / t1 / 1e9:5.0f} dense), rel diff {err:.2e}, "
              f"reproducible {bool(torch.equal(new, new2))}", flush=True)
