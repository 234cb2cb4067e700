import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch, ctypes as C
from treelearn_amd import _hip
from treelearn_amd.cluster import hdbscan
rng = np.random.default_rng(1)
def both(xy, m):
    lg, (gs, gd, gw) = hdbscan(xy, m, algorithm="grid", return_mst=True)
    lp, (ps, pd, pw) = hdbscan(xy, m, algorithm="prim", return_mst=True)
    return np.array_equal(np.sort(gw), np.sort(pw)), np.array_equal(lg, lp), float((lg != lp).mean()), int(lg.max() + 1), int(lp.max() + 1)
cases = {}
g = np.stack(np.meshgrid(np.arange(60), np.arange(60)), -1).reshape(-1, 2).astype(np.float32) * 0.1
cases["lattice"] = (g, 10)
c = rng.uniform(0, 30, (12, 2)); lat = (np.round((c[rng.integers(0, 12, 6000)] + rng.normal(0, 0.3, (6000, 2))) * 20) / 20).astype(np.float32)
cases["quantised blobs (0.05 m lattice, duplicates)"] = (lat, 50)
cases["all identical"] = (np.ones((500, 2), np.float32) * 3.5, 5)
cases["two points"] = (np.array([[0, 0], [1, 1]], np.float32), 2)
cases["line"] = (np.stack([np.linspace(0, 50, 4000), np.zeros(4000)], 1).astype(np.float32), 20)
cases["n == m"] = (rng.normal(size=(50, 2)).astype(np.float32), 50)
cases["far outliers"] = (np.concatenate([rng.normal(0, 0.1, (3000, 2)), rng.normal(0, 0.1, (3000, 2)) + 5, [[1e4, 1e4], [-1e4, 3e3]]]).astype(np.float32), 50)
cases["uniform"] = (rng.uniform(0, 10, (8000, 2)).astype(np.float32), 50)
for k, (xy, m) in cases.items():
    try: print(k, both(xy, m), flush=True)
    except Exception as e: print(k, "EXC", e, flush=True)
from sklearn.cluster import HDBSCAN
for seed in range(3):
    r = np.random.default_rng(seed); c = r.uniform(0, 40, (10, 2)); xy = (c[r.integers(0, 10, 5000)] + r.normal(0, 0.2, (5000, 2))).astype(np.float32)
    sk = HDBSCAN(min_cluster_size=50).fit(xy.astype(np.float64)).labels_
    for alg in ("prim", "grid"):
        l = hdbscan(xy, 50, algorithm=alg); print("sklearn vs", alg, "identical", np.array_equal(sk, l), (sk != l).mean(), flush=True)
