"""Developer A/B of the level-1 down / inverse convs on the config-2 tile: kernel family switched by tl_set_tuning.   python tools/dev_updown.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import _hip, geometry as G, ops
from treelearn_amd.synth import CONFIGS, make_batch, make_tile


def timeit(f, reps=20, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
c, bi = b["coords"].cuda().float(), b["batch_ids"].cuda().long()
L = _hip.lib()
for blocked in (False, True):
    g = G.build_geometry(c, bi, 1, 0.1, 7, [500, 500, 1000], blocked=blocked)
    l0, l1 = g.levels[0], g.levels[1]
    torch.manual_seed(0)
    x1 = torch.randn(l0.n, 32, device="cuda").bfloat16(); x2 = torch.randn(l1.n, 64, device="cuda").bfloat16()
    wd = ops.pack_weight(torch.randn(64, 2, 2, 2, 32, device="cuda") * 0.1, torch.bfloat16)
    wu = ops.pack_weight(torch.randn(32, 2, 2, 2, 64, device="cuda") * 0.1, torch.bfloat16)
    sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda")
    cat_r = torch.empty(l0.n, 64, device="cuda", dtype=torch.bfloat16); cat_a = torch.empty_like(cat_r)
    o2 = torch.empty(l1.n, 64, device="cuda", dtype=torch.bfloat16)
    down = lambda: ops.conv_fwd(x1, wd, l0.child, l1.n, out2=(o2, sc, sh, True))
    up = lambda: ops.conv_fwd(x2, wu, l0.inv, l0.n, out=cat_r[:, 32:], out2=(cat_a[:, 32:], sc[:32], sh[:32], True), one_hot=True)
    res = {}
    for name, f in (("down 32->64", down), ("inverse 64->32", up)):
        for key, val in (("direct", 0), ("direct", 1)) if name.startswith("down") else (("direct_oh", 0), ("direct_oh", 1)):
            L.tl_set_tuning(key.encode(), val)
            out = f(); r = (out.clone(), cat_a.clone())
            t = timeit(f)
            print(f"blocked={blocked} {name}: {key}={val}: {t:.3f} ms", flush=True)
            if val == 0: res[name] = r
            else: print("   identical:", torch.equal(res[name][0], r[0]) and torch.equal(res[name][1], r[1]))
        L.tl_set_tuning(b"direct", 1); L.tl_set_tuning(b"direct_oh", 1)
