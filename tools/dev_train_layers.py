"""Dev: per-layer time of the three conv passes of a config-3 training step (forward, dgrad = forward over the transposed table,
wgrad) on the real rulebooks of a 2-crop batch, bf16.      python tools/dev_train_layers.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import geometry as G, ops
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

tiles = [make_tile(**CONFIGS["config2"], seed=s) for s in (0, 1)]
b = make_batch(tiles)
geom = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 2, 0.1, 7, [500, 500, 1000])
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = [0.0, 0.0, 0.0, 0.0]
print(f"{'layer':28s} {'rows':>9s} {'pairs/row':>9s} {'fwd ms':>8s} {'dgrad':>8s} {'wgrad':>8s} {'TF/s f':>7s} {'d':>6s} {'w':>6s}")
only = set(sys.argv[1:])
for li, lv in enumerate(geom.levels):
    if only and f"l{li+1}" not in only: continue
    C = 32 * (li + 1)
    shapes = [(C, C, "subm", 7 if li < 6 else 4), (2 * C, C, "subm", 1 if li < 6 else 0)]
    if li < 6 and lv is not geom.levels[-1]: shapes += [(C, C + 32, "down", 1), (C + 32, C, "up", 1)]
    for ci, co, kind, mult in shapes:
        if mult == 0: continue
        if kind == "subm": table, n_out, n_in, K, ttab = lv.nbr, lv.n, lv.n, 27, lv.nbr
        elif kind == "down": nx = geom.levels[li + 1]; table, n_out, n_in, K, ttab = lv.child, nx.n, lv.n, 8, lv.inv
        else: nx = geom.levels[li + 1]; table, n_out, n_in, K, ttab = lv.inv, lv.n, nx.n, 8, lv.child
        if table is None or ttab is None: continue
        x = torch.randn(n_in, ci, device="cuda").bfloat16(); g = torch.randn(n_out, co, device="cuda").bfloat16()
        w = (torch.randn(K, co, ci, device="cuda") * 0.05).bfloat16(); wt = w.permute(0, 2, 1).contiguous()
        pairs = int((table >= 0).sum())
        tf = timeit(lambda: ops.conv_fwd(x, w, table, n_out))
        step = None
        if ci <= 224: td = timeit(lambda: ops.conv_fwd(g, wt, ttab, n_in))
        else:
            step = 128 if ci % 128 == 0 else (96 if ci % 96 == 0 else 32)
            gx = torch.empty(n_in, ci, device="cuda", dtype=torch.bfloat16)
            ws_ = [wt[:, s:s + step].contiguous() for s in range(0, ci, step)]
            def dg():
                for i, s in enumerate(range(0, ci, step)): ops.conv_fwd(g, ws_[i], ttab, n_in, out=gx[:, s:s + step])
            td = timeit(dg)
        tw = timeit(lambda: ops.conv_wgrad(x, g, table, n_out, K))
        fl = 2.0 * pairs * ci * co / 1e9
        print(f"l{li+1} {kind:4s} {ci:3d}->{co:3d} x{mult}        {n_out:9d} {pairs / n_out:9.2f} {tf:8.3f} {td:8.3f} {tw:8.3f} {fl / tf:7.0f} {fl / td:6.0f} {fl / tw:6.0f}", flush=True)
        tot[0] += tf * mult; tot[1] += td * mult; tot[2] += tw * mult
print(f"sum over the step's layers: fwd {tot[0]:.1f} ms, dgrad {tot[1]:.1f} ms, wgrad {tot[2]:.1f} ms")
