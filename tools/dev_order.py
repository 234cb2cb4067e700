"""Dev experiment: would a tiled (x, y) voxel order speed the big SubM convs up?  Take the real level-1 / level-2 rulebooks
of config 2, permute the rows into tile-major column order (z stays fastest), and time the same conv on the permuted
problem (the conv result is a permutation of the original one; only the memory access pattern changes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import ops
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_tile

cfg = CONFIGS["config2"]
t = make_tile(**cfg, seed=0)
pts = torch.from_numpy(t["points"]).cuda(); bid = torch.zeros(len(pts), dtype=torch.int64, device="cuda")
g = build_geometry(pts, bid, 1, cfg["voxel"], 7, [500, 500, 1000])


def bench(x, w, nbr, n, res):
    out = torch.empty_like(res)
    for _ in range(5): ops.conv_fwd(x, w, nbr, n, out=out, residual=res)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv_fwd(x, w, nbr, n, out=out, residual=res)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20, out


for level in (0, 1):
    lv = g.levels[level]; C = 32 * (level + 1); n = lv.n
    x = torch.randn(n, C, device="cuda").to(torch.bfloat16); res = torch.randn(n, C, device="cuda").to(torch.bfloat16)
    w = ops.pack_weight(torch.randn(C, 3, 3, 3, C, device="cuda") * 0.05, torch.bfloat16)
    nbr = lv.nbr.clone()                                  # plain table (no column form attached)
    ms0, ref = bench(x, w, nbr, n, res)
    print(f"level {level + 1} C={C} n={n}: original (x, y, z) order {ms0:.3f} ms")
    c = lv.coords.long()
    for name, T in (("4x4 column tiles", 4), ("8x8", 8), ("16x16", 16), ("32x32", 32)):
        key = (((c[:, 1] // T) * 4096 + (c[:, 2] // T)) * T + (c[:, 1] % T)) * T + (c[:, 2] % T)
        key = key * 4096 + c[:, 3]
        perm = torch.argsort(key)                          # new row r' holds old row perm[r']
        inv = torch.empty_like(perm); inv[perm] = torch.arange(n, device="cuda")
        nb2 = nbr[:, perm].long()
        nb2 = torch.where(nb2 >= 0, inv[nb2.clamp(min=0)], nb2).to(torch.int32).contiguous()
        ms, out = bench(x[perm].contiguous(), w, nb2, n, res[perm].contiguous())
        ok = torch.equal(out, ref[perm])
        print(f"   {name:18s} {ms:.3f} ms  ({ms0 / ms:.2f}x)  same result: {ok}")
