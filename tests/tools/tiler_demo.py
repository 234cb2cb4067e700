"""Measure the device tiler on a synthetic plot: tiles/s and points/s of PlotTiler with and without the
host offset-label derivation, beside the CPU restatement of the reference chain (oracle/tiles.py; no npz I/O, which the
reference additionally pays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from treelearn_amd.synth import make_tile
from treelearn_amd.util.tiles import PlotTiler

extent = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t = make_tile(extent=extent, voxel=0.1, n_trees=int(64 * (extent / 40) ** 2), fill=0.10, seed=3)
pts = t["points"].astype(np.float32); labels = t["instance_label"].astype(np.float32); feats = t["feat"].astype(np.float32).reshape(len(pts), -1)
print(f"plot {extent:.0f} x {extent:.0f} m, {len(pts)} points")
tiler = PlotTiler(pts, labels, feats)
for mode in ("none", "host"):
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0; npts = 0
    for b in tiler.tiles(8.0, 13.5, 0.5, 8.0, offset_labels=mode):
        n += 1; npts += len(b["coords"])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"PlotTiler offset_labels={mode}: {n} tiles, {npts / n:.0f} points/tile, {dt / n * 1e3:.2f} ms/tile, {npts / dt / 1e6:.1f} Mpoints/s")
if os.environ.get("TL_TILER_CPU", "1") == "1":
    from oracle import tiles as ot
    t0 = time.perf_counter(); res = ot.plot_tiles(pts, labels, feats, 8.0, 13.5, 0.5, 8.0); dt = time.perf_counter() - t0
    print(f"CPU restatement of the reference chain (1 core, no npz I/O): {len(res)} tiles, {dt / len(res) * 1e3:.1f} ms/tile")
