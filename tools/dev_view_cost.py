"""What an extra output view of a level-2 conv costs, and whether it is the bytes or the stores: the 64 -> 64 SubM conv of the config-2 tile
with one / two / three views, with the extra views ALIASED onto the first one (same stores, no additional memory traffic), and with the views
written as halves of a concat buffer (row pitch 256 B instead of 128).   python tools/dev_view_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import geometry as G, ops
from treelearn_amd.synth import CONFIGS, make_batch, make_tile


def timeit(f, reps=30, warm=5):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if os.environ.get("VIEW_Q"):                              # dev build: a variant of the stream-q kernel for the 64 -> 64 shape (tl_conv_streamq.hip)
    from treelearn_amd import _hip
    _hip.lib().tl_dev_streamq_tm(int(os.environ["VIEW_Q"]), None)
    print("stream-q variant", os.environ["VIEW_Q"])
b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
g = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 1, 0.1, 7, [500, 500, 1000])
for li, C in ((1, 64), (2, 96)):
    lv = g.levels[li]; n = lv.n
    torch.manual_seed(0)
    x = torch.randn(n, C, device="cuda").bfloat16(); res = torch.randn(n, C, device="cuda").bfloat16()
    w = ops.pack_weight(torch.randn(C, 3, 3, 3, C, device="cuda") * 0.05, torch.bfloat16)
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    o1, o2, o3 = (torch.empty(n, C, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    cat1, cat2 = (torch.empty(n, 2 * C, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    cases = [
        ("1 view", dict(out=o1)),
        ("2 views", dict(out=o1, out2=(o2, sc, sh, True))),
        ("3 views", dict(out=o1, out2=(o2, sc, sh, True), out3=(o3, sc, sh, True))),
        ("2 views, second aliased onto the first", dict(out=o1, out2=(o1, None, None, False))),
        ("3 views, both aliased onto the first", dict(out=o1, out2=(o1, None, None, False), out3=(o1, None, None, False))),
        ("1 view into a concat half (pitch 2C)", dict(out=cat1[:, :C])),
        ("2 views into concat halves", dict(out=cat1[:, :C], out2=(cat2[:, :C], sc, sh, True))),
    ]
    best = {name: [] for name, _ in cases}
    for rnd in range(5):                                   # interleaved rounds: the part's clock settles over the first seconds
        for name, kw in cases:
            best[name].append(timeit(lambda: ops.conv_fwd(x, w, lv.nbr, n, residual=res, **kw), reps=15, warm=2))
    for name, _ in cases:
        v = sorted(best[name][1:])
        print("level %d %d->%d rows %d  %-44s median %.3f ms  (min %.3f max %.3f)" % (li + 1, C, C, n, name, v[len(v) // 2], v[0], v[-1]), flush=True)
