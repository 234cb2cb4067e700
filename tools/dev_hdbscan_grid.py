"""Dev: the quadtree / Boruvka HDBSCAN device stage against the Prim form (core distances bit-equal, MST weight multiset equal,
labels compared) and its time at scale.     python tools/dev_hdbscan_grid.py [n ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from treelearn_amd import _hip
from treelearn_amd.cluster import hdbscan


def blobs(n, rng, spread=0.15, extent=100.0, noise=0.05):
    k = max(n // 2500, 4)
    centers = rng.uniform(0, extent, size=(k, 2))
    xy = (centers[rng.integers(0, k, n)] + rng.normal(0, spread, size=(n, 2))).astype(np.float32)
    m = int(n * noise)
    xy[:m] = rng.uniform(0, extent, size=(m, 2)).astype(np.float32)
    return xy


def disagreement(a, b):
    """fraction of points whose label differs after renaming every cluster of b to the cluster of a it overlaps most; noise mismatches"""
    ren = {-1: -1}
    for c in np.unique(b[b >= 0]):
        v, cnt = np.unique(a[b == c], return_counts=True)
        ren[c] = v[np.argmax(cnt)]
    br = np.array([ren[c] for c in b])
    return float((br != a).mean()), int(((a < 0) != (b < 0)).sum())


def partition_equal(a, b):
    """same partition up to renaming (noise = -1 must match exactly)"""
    if not np.array_equal(a < 0, b < 0): return False
    m = a >= 0
    pairs = np.unique(np.stack([a[m], b[m]], 1), axis=0)
    return len(pairs) == len(np.unique(a[m])) == len(np.unique(b[m]))


rng = np.random.default_rng(0)
hdbscan(blobs(3000, rng), 50, algorithm="grid")
for n in [int(a) for a in sys.argv[1:]] or [3000, 20000, 60000, 150000, 400000, 1000000]:
    xy = blobs(n, rng)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lg, (gs, gd, gw) = hdbscan(xy, 50, algorithm="grid", return_mst=True)
    torch.cuda.synchronize(); tg = time.perf_counter() - t0
    msg = f"n={n}: grid {tg:.3f} s, clusters {lg.max() + 1}, noise {(lg < 0).mean():.3f}"
    if n <= 160000:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lp, (ps, pd, pw) = hdbscan(xy, 50, algorithm="prim", return_mst=True)
        torch.cuda.synchronize(); tp = time.perf_counter() - t0
        msg += (f" | prim {tp:.3f} s | weights equal: {np.array_equal(np.sort(gw), np.sort(pw))}, labels identical: {np.array_equal(lg, lp)}, "
                f"same partition: {partition_equal(lg, lp)}, clusters prim {lp.max() + 1}, (disagreement after renaming, noise mismatches) = {disagreement(lp, lg)}")
    print(msg, flush=True)
    # device stage alone
    L = _hip.lib(); t = torch.from_numpy(xy).cuda()
    import ctypes as C
    grid = _hip.HdbGrid(); pws = torch.empty(int(L.tl_hdbscan_grid_plan_ws_bytes()), dtype=torch.uint8, device="cuda")
    es = torch.empty(n - 1, dtype=torch.int32, device="cuda"); ed = torch.empty_like(es); ew = torch.empty(n - 1, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _hip.check(L.tl_hdbscan_grid_plan(_hip.ptr(t), n, C.addressof(grid), _hip.ptr(pws), _hip.stream()), "plan")
    ws = torch.empty(int(L.tl_hdbscan_grid_ws_bytes(n, C.addressof(grid))), dtype=torch.uint8, device="cuda")
    _hip.check(L.tl_hdbscan_mst_grid(_hip.ptr(t), n, 50, C.addressof(grid), _hip.ptr(es), _hip.ptr(ed), _hip.ptr(ew), None, _hip.ptr(ws), _hip.stream()), "mst")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"   device stage {dt:.3f} s; leaf grid 2^{grid.levels}, h = {grid.h:.4f}, ws {ws.numel() / 1e6:.0f} MB", flush=True)
