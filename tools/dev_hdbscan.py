"""Dev: time the device part of HDBSCAN (tl_hdbscan_mst: core distances + Prim MST, one launch per MST edge) and the host
labelling on trunk-like blobs; for n <= 5000 the edge list is checked against a brute-force numpy Prim with the same rule.

    python tools/dev_hdbscan.py [n ...]
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from treelearn_amd import _hip
from treelearn_amd.cluster import hdbscan

rng = np.random.default_rng(0)
for n in [int(a) for a in sys.argv[1:]] or [3000, 20000, 50000, 150000]:
    k = max(n // 2500, 4)
    centers = rng.uniform(0, 100, size=(k, 2))
    xy = (centers[rng.integers(0, k, n)] + rng.normal(0, 0.15, size=(n, 2))).astype(np.float32)
    xy[: n // 20] = rng.uniform(0, 100, size=(n // 20, 2)).astype(np.float32)          # scattered noise
    hdbscan(xy[:2000], 50)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lab = hdbscan(xy, 50)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"n={n}: {dt:.3f} s  ({dt / n * 1e6:.2f} us per point)  clusters={lab.max() + 1} noise={(lab < 0).mean():.3f}", flush=True)
    L = _hip.lib(); t = torch.from_numpy(xy).cuda()
    es = torch.empty(n - 1, dtype=torch.int32, device="cuda"); ed = torch.empty_like(es); ew = torch.empty(n - 1, dtype=torch.float64, device="cuda")
    ws = torch.empty(int(L.tl_hdbscan_ws_bytes(n)), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _hip.check(L.tl_hdbscan_mst(_hip.ptr(t), n, 50, _hip.ptr(es), _hip.ptr(ed), _hip.ptr(ew), None, _hip.ptr(ws), _hip.stream()), "mst")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"   device stage: {dt:.3f} s = {dt / n * 1e6:.2f} us per point")
    if n <= 5000:                                          # brute-force Prim with the same rule (smallest index among equal reachabilities)
        X = xy.astype(np.float64); D = np.sqrt(((X[:, None, :] - X[None, :, :]) ** 2).sum(-1))
        corev = np.sort(D, axis=1)[:, 49]
        reach = np.full(n, np.inf); intree = np.zeros(n, bool); intree[0] = True; cur = 0; dst = []
        for step in range(n - 1):
            mr = np.maximum(np.maximum(corev[cur], corev), D[cur]); upd = (mr < reach) & ~intree
            reach[upd] = mr[upd]
            cand = np.where(intree, np.inf, reach); j = int(np.argmin(cand))        # argmin = first minimum = smallest index
            dst.append(j); intree[j] = True; cur = j
        print(f"   edges vs numpy Prim: {int((ed.cpu().numpy() != np.array(dst)).sum())} differ")
