"""Instance grouping on the GPU (reference tree_learn/util/pipeline.py:173-191)."""
import ctypes as _c
import os

import numpy as np
import torch

from . import _hip


def dbscan_min2(xy, eps, device="cuda"):
    """sklearn DBSCAN(eps, min_samples=2).fit(xy).labels_ on the HIP library: int64 labels, -1 = noise,
    clusters numbered by their smallest point index.  `xy`: float32 [n,2] numpy array or tensor."""
    L = _hip.lib()
    t = torch.as_tensor(np.ascontiguousarray(xy, dtype=np.float32)) if not torch.is_tensor(xy) else xy.float().contiguous()
    n = t.shape[0]
    if n == 0:
        return np.zeros(0, np.int64)
    t = t.to(device)
    labels = torch.empty(n, dtype=torch.int32, device=t.device)
    ncl = torch.empty(1, dtype=torch.int32, device=t.device)
    ws = torch.empty(int(L.tl_cluster_ws_bytes(n)), dtype=torch.uint8, device=t.device)
    _hip.check(L.tl_cluster_grid(_hip.ptr(t), n, float(eps), _hip.ptr(labels), _hip.ptr(ncl), _hip.ptr(ws), _hip.stream()), "tl_cluster_grid")
    return labels.cpu().numpy().astype(np.int64)


MAX_MIN_SAMPLES = 4096         # grid form: beyond 128 neighbours the k-best lists are heaps in the workspace (csrc/tl_hdbscan_grid.hip k_core_big)
PRIM_MAX_MIN_SAMPLES = 128     # kMaxK of csrc/tl_hdbscan.hip: the Prim form keeps the k-best list in registers / scratch
# "auto": above this share of exactly tied MST weights the grid form's tree is re-built in Prim's order.  Mutual reachability alone ties
# 1-2 % of the weights on smooth data (an edge often weighs some point's core distance: 142 of 11 999 at min_samples 50) with labels
# identical to the Prim form; the inputs on which the two forms split a level differently had 19-100 % ties (tools/fuzz_hdbscan.py)
TIE_FRACTION = 0.10
# ... but only up to this many points: the Prim form is O(n^2) (50 k points: ~1 s; 400 k: minutes).  Above it the grid form's tree is kept
# (it is a minimal tree of the same weights, put into Prim's order; labels can differ from sklearn's only where exact ties decide a split)
# and a warning says so.
PRIM_FALLBACK_MAX_POINTS = int(os.environ.get("TL_HDBSCAN_PRIM_MAX", "50000"))     # (argument `prim_fallback_max` / env TL_HDBSCAN_PRIM_MAX)


GRID_MIN_POINTS = 8192          # from here on the quadtree / Boruvka device stage replaces the two O(n^2) passes


def hdbscan(xy, min_cluster_size, device="cuda", return_mst=False, algorithm="auto", prim_fallback_max=None):
    """sklearn HDBSCAN(min_cluster_size=m).fit(xy).labels_ (min_samples = m, EOM): core distances + MST of the mutual-reachability
    graph on the GPU, hierarchy condensation on the host (tl_hdbscan_labels_host).
    algorithm: "prim" = tl_hdbscan_mst (O(n^2), sklearn's edge order exactly), "grid" = tl_hdbscan_mst_grid (quadtree k-NN + Boruvka,
    same core distances and MST weights, its own deterministic rule among equal-weight edges), "auto" = grid from GRID_MIN_POINTS points.
    return_mst=True also returns the MST as (src, dst, weight) numpy arrays in Prim order."""
    L = _hip.lib()
    t = torch.as_tensor(np.ascontiguousarray(xy, dtype=np.float32)) if not torch.is_tensor(xy) else xy.float().contiguous()
    n = t.shape[0]
    m = int(min_cluster_size)
    if n < m:
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {m}, n_samples_fit = {n}")   # as sklearn
    if m > MAX_MIN_SAMPLES:
        raise ValueError(f"min_cluster_size = {m} exceeds the {MAX_MIN_SAMPLES} neighbours the HIP core-distance kernels keep per point "
                         f"(tau_min of the reference's grouping config is 50)")
    if algorithm not in ("auto", "prim", "grid"):
        raise ValueError(f"unknown HDBSCAN algorithm {algorithm!r}")
    if algorithm == "prim" and m > PRIM_MAX_MIN_SAMPLES:
        raise ValueError(f"the Prim form keeps at most {PRIM_MAX_MIN_SAMPLES} neighbours per point; use algorithm='grid' / 'auto' for min_cluster_size = {m}")
    t = t.to(device)
    e_src = torch.empty(n - 1, dtype=torch.int32, device=t.device)
    e_dst = torch.empty(n - 1, dtype=torch.int32, device=t.device)
    e_w = torch.empty(n - 1, dtype=torch.float64, device=t.device)
    use_grid = algorithm == "grid" or (algorithm == "auto" and (n >= GRID_MIN_POINTS or m > PRIM_MAX_MIN_SAMPLES))
    if use_grid:
        if not bool(torch.isfinite(t).all()):
            raise ValueError("hdbscan: non-finite coordinates")
        grid = _hip.HdbGrid()
        pws = torch.empty(int(L.tl_hdbscan_grid_plan_ws_bytes()), dtype=torch.uint8, device=t.device)
        _hip.check(L.tl_hdbscan_grid_plan(_hip.ptr(t), n, _c.addressof(grid), _hip.ptr(pws), _hip.stream()), "tl_hdbscan_grid_plan")
        ws = torch.empty(int(L.tl_hdbscan_grid_ws_bytes_k(n, _c.addressof(grid), m)), dtype=torch.uint8, device=t.device)
        _hip.check(L.tl_hdbscan_mst_grid(_hip.ptr(t), n, m, _c.addressof(grid), _hip.ptr(e_src), _hip.ptr(e_dst), _hip.ptr(e_w), None, _hip.ptr(ws),
                                         _hip.stream()), "tl_hdbscan_mst_grid")
        gs, gd, gw = e_src.cpu().numpy(), e_dst.cpu().numpy(), e_w.cpu().numpy()
        if algorithm == "auto" and m <= PRIM_MAX_MIN_SAMPLES:
            # Among equal-weight edges the grid form picks by (weight, indices), Prim / sklearn by insertion order: any of those trees is
            # minimal, but the condensed hierarchy may split a tied level differently.  Where ties are common (quantised or duplicated
            # coordinates, tiny min_samples) the default therefore re-builds the tree in Prim's own order -- slower, sklearn's labels.
            ws_ = np.sort(gw)
            if (ws_[1:] == ws_[:-1]).sum() > TIE_FRACTION * max(len(gw), 1):
                if n <= (PRIM_FALLBACK_MAX_POINTS if prim_fallback_max is None else prim_fallback_max):
                    return hdbscan(xy, min_cluster_size, device=device, return_mst=return_mst, algorithm="prim")
                import warnings
                warnings.warn(f"hdbscan: {n} points with more than {TIE_FRACTION:.0%} exactly tied tree weights (quantised or duplicated "
                              f"coordinates?); above {PRIM_FALLBACK_MAX_POINTS} points the quadtree form's tree is kept -- labels may differ "
                              f"from the O(n^2) form's where ties decide a split (algorithm='prim' forces it)", RuntimeWarning, stacklevel=2)
        hs, hd, hw = np.empty_like(gs), np.empty_like(gd), np.empty_like(gw)       # the tree in Prim's order / orientation (host)
        _hip.check(L.tl_hdbscan_prim_order_host(gs.ctypes.data, gd.ctypes.data, gw.ctypes.data, n, hs.ctypes.data, hd.ctypes.data, hw.ctypes.data),
                   "tl_hdbscan_prim_order_host")
    else:
        ws = torch.empty(int(L.tl_hdbscan_ws_bytes(n)), dtype=torch.uint8, device=t.device)
        _hip.check(L.tl_hdbscan_mst(_hip.ptr(t), n, m, _hip.ptr(e_src), _hip.ptr(e_dst), _hip.ptr(e_w), None, _hip.ptr(ws), _hip.stream()), "tl_hdbscan_mst")
        hs, hd, hw = e_src.cpu().numpy(), e_dst.cpu().numpy(), e_w.cpu().numpy()
    labels = np.empty(n, np.int32)
    _hip.check(L.tl_hdbscan_labels_host(hs.ctypes.data, hd.ctypes.data, hw.ctypes.data, n, m, labels.ctypes.data), "tl_hdbscan_labels_host")
    return (labels.astype(np.int64), (hs, hd, hw)) if return_mst else labels.astype(np.int64)
