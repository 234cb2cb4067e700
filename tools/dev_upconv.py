"""Dev: what the inverse (up) convs would cost with ONE tap's matrix work per row (K = 1 conv through the parent table) against the
shipped one-hot 8-tap form, and the down convs against a K = 1 conv of the same rows (lower bound of their traffic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import geometry as G, ops
from treelearn_amd.synth import CONFIGS, make_batch, make_tile
b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
geom = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 1, 0.1, 7, [500, 500, 1000])
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for li in range(3):
    lv, nx = geom.levels[li], geom.levels[li + 1]
    C, C2 = 32 * (li + 1), 32 * (li + 2)
    # up: C2 -> C over lv.inv (one entry per row)
    x = torch.randn(nx.n, C2, device="cuda").bfloat16()
    w8 = ops.pack_weight(torch.randn(C, 2, 2, 2, C2, device="cuda") * 0.05, torch.bfloat16)
    w1 = ops.pack_weight(torch.randn(C, 1, 1, 1, C2, device="cuda") * 0.05, torch.bfloat16)
    parent = lv.inv.max(0).values.reshape(1, -1).contiguous()
    out = torch.empty(lv.n, C, device="cuda", dtype=torch.bfloat16); out2 = torch.empty_like(out)
    s2 = torch.ones(C, device="cuda"); h2 = torch.zeros(C, device="cuda")
    t8 = timeit(lambda: ops.conv_fwd(x, w8, lv.inv, lv.n, out=out, out2=(out2, s2, h2, True), one_hot=True))
    t1 = timeit(lambda: ops.conv_fwd(x, w1, parent, lv.n, out=out, out2=(out2, s2, h2, True)))
    print(f"l{li+1} up   {C2:3d}->{C:3d} rows {lv.n:8d}: one-hot 8-tap {t8:.3f} ms, one tap through the parent table {t1:.3f} ms", flush=True)
    # down: C -> C2 over nx.child
    xd = torch.randn(lv.n, C, device="cuda").bfloat16()
    wd = ops.pack_weight(torch.randn(C2, 2, 2, 2, C, device="cuda") * 0.05, torch.bfloat16)
    wd1 = ops.pack_weight(torch.randn(C2, 1, 1, 1, C, device="cuda") * 0.05, torch.bfloat16)
    od = torch.empty(nx.n, C2, device="cuda", dtype=torch.bfloat16); od2 = torch.empty_like(od)
    s3 = torch.ones(C2, device="cuda"); h3 = torch.zeros(C2, device="cuda")
    first = lv.child.max(0).values.reshape(1, -1).contiguous()
    td = timeit(lambda: ops.conv_fwd(xd, wd, lv.child, nx.n, out=od, out2=(od2, s3, h3, True)))
    td1 = timeit(lambda: ops.conv_fwd(xd, wd1, first, nx.n, out=od, out2=(od2, s3, h3, True)))
    print(f"l{li+1} down {C:3d}->{C2:3d} rows {nx.n:8d}: 8-tap {td:.3f} ms, one tap {td1:.3f} ms", flush=True)
