import sys, os
sys.path.insert(0, os.getcwd())
import torch
from treelearn_amd import geometry as G
from treelearn_amd.synth import CONFIGS, make_batch, make_tile
tile = make_tile(**CONFIGS["config2"], seed=0)
b = make_batch([tile])
geom = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 1, 0.1, 7, [500, 500, 1000])
for li in range(4):
    nbr = geom.levels[li].nbr
    n = nbr.shape[1]
    for R in (16, 32, 64):
        m = (n // R) * R
        pres = (nbr[:, :m] >= 0).view(27, m // R, R)
        anyp = pres.any(dim=2)
        cnt = pres.sum(dim=2).float()
        print(f"level {li+1} rows {n} pairs/row {float((nbr>=0).sum())/n:.2f}: tiles of {R}: non-empty (tile, tap) {float(anyp.float().mean())*100:.1f} %, mean present rows in a non-empty one {float(cnt[anyp].mean()):.1f}")
