import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch, ctypes as C
from treelearn_amd import _hip
from treelearn_amd.cluster import hdbscan
rng = np.random.default_rng(0)
L = _hip.lib()
for name, xy in (("uniform 1.5M", rng.uniform(0, 60, (1500000, 2)).astype(np.float32)),
                 ("quantised 0.01 grid 1.5M", (np.round(rng.uniform(0, 60, (1500000, 2)) * 100) / 100).astype(np.float32)),
                 ("surface-like (blobs + uniform) 1.5M", np.concatenate([rng.uniform(0, 60, (750000, 2)), rng.uniform(0, 60, (300, 2))[rng.integers(0, 300, 750000)] + rng.normal(0, 0.2, (750000, 2))]).astype(np.float32))):
    n = len(xy); t = torch.from_numpy(xy).cuda()
    hdbscan(xy[:5000], 50, algorithm="grid")
    grid = _hip.HdbGrid(); pws = torch.empty(int(L.tl_hdbscan_grid_plan_ws_bytes()), dtype=torch.uint8, device="cuda")
    es = torch.empty(n - 1, dtype=torch.int32, device="cuda"); ed = torch.empty_like(es); ew = torch.empty(n - 1, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _hip.check(L.tl_hdbscan_grid_plan(_hip.ptr(t), n, C.addressof(grid), _hip.ptr(pws), _hip.stream()), "plan")
    ws = torch.empty(int(L.tl_hdbscan_grid_ws_bytes(n, C.addressof(grid))), dtype=torch.uint8, device="cuda")
    _hip.check(L.tl_hdbscan_mst_grid(_hip.ptr(t), n, 50, C.addressof(grid), _hip.ptr(es), _hip.ptr(ed), _hip.ptr(ew), None, _hip.ptr(ws), _hip.stream()), "mst")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    gs, gd, gw = es.cpu().numpy(), ed.cpu().numpy(), ew.cpu().numpy()
    hs, hd, hw = np.empty_like(gs), np.empty_like(gd), np.empty_like(gw)
    t2 = time.perf_counter()
    L.tl_hdbscan_prim_order_host(gs.ctypes.data, gd.ctypes.data, gw.ctypes.data, n, hs.ctypes.data, hd.ctypes.data, hw.ctypes.data)
    t3 = time.perf_counter()
    lab = np.empty(n, np.int32)
    L.tl_hdbscan_labels_host(hs.ctypes.data, hd.ctypes.data, hw.ctypes.data, n, 50, lab.ctypes.data)
    t4 = time.perf_counter()
    print(f"{name}: device {t1 - t0:.3f} s (grid 2^{grid.levels}), D2H {t2 - t1:.3f}, prim order {t3 - t2:.3f}, labels {t4 - t3:.3f}; clusters {lab.max() + 1}", flush=True)
