"""Dev (needs `python -m treelearn_amd.build --dev`): wgrad variants per layer shape on config-3 rulebooks.
mode = 1 | (2 = per-tap kernel) | variant << 8 | chunk selector << 12"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import geometry as G, ops, _hip
from treelearn_amd.synth import CONFIGS, make_batch, make_tile
b = make_batch([make_tile(**CONFIGS["config2"], seed=s) for s in (0, 1)])
geom = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 2, 0.1, 7, [500, 500, 1000])
L = _hip.lib()
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
modes = [("per-tap", 3), ("shared", 1)] + [(f"v{v}", 1 | (v << 8)) for v in range(1, 8)]
print("layer".ljust(22), " ".join(n.rjust(7) for n, _ in modes))
for li in range(0, 5):
    lv = geom.levels[li]; C = 32 * (li + 1)
    shapes = [(C, C, "subm"), (2 * C, C, "subm"), (C, C + 32, "down"), (C + 32, C, "up")]
    for ci, co, kind in shapes:
        if kind == "subm": table, n_out, n_in, K = lv.nbr, lv.n, lv.n, 27
        elif kind == "down": nx = geom.levels[li + 1]; table, n_out, n_in, K = lv.child, nx.n, lv.n, 8
        else: nx = geom.levels[li + 1]; table, n_out, n_in, K = lv.inv, lv.n, nx.n, 8
        x = torch.randn(n_in, ci, device="cuda").bfloat16(); g = torch.randn(n_out, co, device="cuda").bfloat16()
        ref = None; row = []
        for name, m in modes:
            L.tl_dev_wgrad_mode(m)
            r = ops.conv_wgrad(x, g, table, n_out, K)
            if ref is None: ref = r
            else: assert (r - ref).abs().max() <= 1e-3 * ref.abs().max(), (name, float((r - ref).abs().max()), float(ref.abs().max()))
            row.append(timeit(lambda: ops.conv_wgrad(x, g, table, n_out, K)))
        print(f"l{li+1} {kind:4s} {ci:3d}->{co:3d}".ljust(22), " ".join(f"{t:7.3f}" for t in row), flush=True)
