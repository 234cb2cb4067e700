"""Dev (plan of DESIGN.md R3.6): statistics of a block-local row order on the real config-2 rulebooks.  Voxels are sorted by (batch, B^3 block key,
reference order inside the block) and cut into units of <= R consecutive rows of one block; per unit: own rows, halo rows (distinct neighbour rows
outside the unit), staged rows = own + halo.      python tools/dev_blocked_stats.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import geometry as G
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

tile = make_tile(**CONFIGS["config2"], seed=0)
b = make_batch([tile])
geom = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 1, 0.1, 7, [500, 500, 1000])


def blocked(lv, B, R):
    c = lv.coords.long()                                    # [n, 4] = (batch, x, y, z)
    n = c.shape[0]
    key = ((c[:, 0] * 4096 + (c[:, 1] // B)) * 4096 + (c[:, 2] // B)) * 4096 + (c[:, 3] // B)
    perm = torch.argsort(key, stable=True)                  # new row j holds old row perm[j]
    o2n = torch.empty_like(perm); o2n[perm] = torch.arange(n, device=perm.device)
    ks = key[perm]
    first = torch.ones(n, dtype=torch.bool, device=perm.device); first[1:] = ks[1:] != ks[:-1]
    bid = torch.cumsum(first.long(), 0) - 1                 # block ordinal of every new row
    bstart = torch.nonzero(first).flatten()
    pos_in_block = torch.arange(n, device=perm.device) - bstart[bid]
    unit_first = first | (pos_in_block % R == 0)
    uid = torch.cumsum(unit_first.long(), 0) - 1
    nu = int(uid[-1]) + 1
    own = torch.bincount(uid, minlength=nu)
    nbr = lv.nbr.long()[:, perm]                            # [27, n] neighbours (old rows) of the new rows
    pres = nbr >= 0
    nn = torch.where(pres, o2n[nbr.clamp(min=0)], torch.full_like(nbr, -1))   # neighbours in new rows
    u_of_n = torch.where(pres, uid[nn.clamp(min=0)], torch.full_like(nn, -1))
    outside = pres & (u_of_n != uid[None, :])
    # distinct (unit, neighbour row) pairs over the outside neighbours
    pairs = (uid[None, :].expand_as(nn)[outside] * n + nn[outside])
    up = torch.unique(pairs)
    halo = torch.bincount(up // n, minlength=nu)
    staged = own + halo
    q = lambda t, p: float(torch.quantile(t.float(), p))
    print(f"  B={B} R={R:3d}: blocks {bstart.numel()}, units {nu}, own rows/unit mean {float(own.float().mean()):.1f}, halo mean {float(halo.float().mean()):.1f} "
          f"(p50 {q(halo, .5):.0f}, p99 {q(halo, .99):.0f}, max {int(halo.max())}), staged p99 {q(staged, .99):.0f} max {int(staged.max())}, "
          f"staged/own over the level {float(staged.sum()) / n:.2f}, 16-row MFMA groups padded/own {float(((own + 15) // 16 * 16).sum()) / n:.2f}")


for li in (0, 1):
    lv = geom.levels[li]
    print(f"level {li + 1}: rows {lv.n}, pairs/row {float((lv.nbr >= 0).sum()) / lv.n:.2f}")
    for B, R in ((8, 64), (8, 128), (4, 64), (16, 64), (16, 128), (8, 32)):
        blocked(lv, B, R)
