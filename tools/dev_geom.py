"""Time build_geometry alone on the config-2 tile: two-call pyramid (default) vs per-level calls (TL_GEOM=per_level).

    python tools/dev_geom.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import make_tile

t = make_tile(extent=40.0, voxel=0.1, n_trees=60, fill=0.10, seed=0)
pts = torch.from_numpy(t["points"]).cuda(); bid = torch.zeros(len(pts), dtype=torch.int64, device="cuda")
print("points", len(pts))
for rep in range(2):
    for mode in ("pyramid", "per_level"):
        if mode == "per_level":
            os.environ["TL_GEOM"] = "per_level"
        else:
            os.environ.pop("TL_GEOM", None)
        for _ in range(5):
            build_geometry(pts, bid, 1, 0.1, 7, [500, 500, 1000])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            g = build_geometry(pts, bid, 1, 0.1, 7, [500, 500, 1000])
        torch.cuda.synchronize()
        print(f"{mode:10s} {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms  n={[lv.n for lv in g.levels]}")
