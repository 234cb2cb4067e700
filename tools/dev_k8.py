"""The strided (k2 s2) and inverse convs of config 2, one at a time: time against what their bytes cost, with the kernel families
switched (tl_set_tuning) and with / without the second output view.  python tools/dev_k8.py > profiles/<tag>/k8_table.txt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import geometry as G, ops, _hip
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
geom = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 1, 0.1, 7, [500, 500, 1000], blocked=True)
L = _hip.lib()
CH = [32, 64, 96, 128, 160]


def t(f, n=30):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def tune(**kw):
    for k, v in kw.items():
        L.tl_set_tuning(k.encode(), v)


print(f"{'conv':28s} {'variant':26s} {'ms':>7s} {'GB (1 view)':>11s} {'GB/s':>6s}")
for li in range(4):
    lv, nx = geom.levels[li], geom.levels[li + 1]
    ci, co = CH[li], CH[li + 1]
    x = torch.randn(lv.n, ci, device="cuda").bfloat16()
    w = ops.pack_weight(torch.randn(co, 2, 2, 2, ci, device="cuda") * 0.1, torch.bfloat16)
    out = torch.empty(nx.n, co, device="cuda", dtype=torch.bfloat16); out2 = torch.empty_like(out)
    sc, sh = torch.rand(co, device="cuda") + 0.5, torch.randn(co, device="cuda")
    gb = (lv.n * ci * 2 + nx.n * co * 2 + 8 * nx.n * 4) / 1e9
    variants = [("default, 2 views", {}, True), ("default, 1 view", {}, False), ("direct=0, 2 views", {"direct": 0}, True),
                ("direct=0 streamq=0, 2 views", {"direct": 0, "streamq": 0}, True)]
    variants.append(("one row block per wave, 2 views", {"stream_rb": 1}, True))
    variants.append(("streamq=0, two row blocks, 2 views", {"direct": 0, "streamq": 0, "stream_rb": 2}, True))
    for name, kw, two in variants:
        tune(**kw)
        f = (lambda: ops.conv_fwd(x, w, lv.child, nx.n, out=out, out2=(out2, sc, sh, True))) if two else (lambda: ops.conv_fwd(x, w, lv.child, nx.n, out=out))
        ms = t(f)
        tune(direct=1, streamq=1, stream_rb=0)
        print(f"down l{li+1}->l{li+2} {ci:3d}->{co:3d} n={nx.n:8d}".ljust(28), f"{name:26s} {ms:7.3f} {gb:11.3f} {gb / ms * 1e3:6.0f}", flush=True)
    # inverse: level li+1 -> li
    x = torch.randn(nx.n, co, device="cuda").bfloat16()
    w = ops.pack_weight(torch.randn(ci, 2, 2, 2, co, device="cuda") * 0.1, torch.bfloat16)
    cat = torch.empty(lv.n, 2 * ci, device="cuda", dtype=torch.bfloat16); act = torch.empty_like(cat)
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda")
    gb = (nx.n * co * 2 + lv.n * ci * 2 + lv.n * 4) / 1e9
    upv = [("default(up), 1 view", {}, False), ("default(up), 2 views", {}, True), ("up=0, 1 view", {"up": 0}, False), ("up=0 direct_oh=0, 1 view", {"up": 0, "direct_oh": 0}, False)]
    for name, kw, two in upv:
        tune(**kw)
        o2 = (act[:, ci:], sc, sh, True) if two else None
        f = lambda: ops.conv_fwd(x, w, lv.inv, lv.n, out=cat[:, ci:], out2=o2, one_hot=True, scatter=lv.child)
        ms = t(f)
        tune(up=1, direct_oh=1)
        print(f"up   l{li+2}->l{li+1} {co:3d}->{ci:3d} n={lv.n:8d}".ljust(28), f"{name:26s} {ms:7.3f} {gb:11.3f} {gb / ms * 1e3:6.0f}", flush=True)
# calibration: what a plain copy of the same bytes costs
for mb in (118, 141, 282):
    a = torch.empty(mb << 20, dtype=torch.uint8, device="cuda"); c = torch.empty_like(a)
    ms = t(lambda: c.copy_(a))
    print(f"copy {mb} MiB: {ms:.3f} ms = {2 * (mb << 20) / ms / 1e6:.0f} GB/s (read + write)")
