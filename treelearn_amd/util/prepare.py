"""Plot preparation on the device -- SURVEY.md 8f #4: what `generate_tiles` does before tiling
(reference tree_learn/util/pipeline.py:24-65) and what the pipeline does after grouping to carry the
predictions back to the original cloud (util/pipeline.py:423-452).

  voxelize           <- data_preparation.py:60-79 (open3d 0.17.0 VoxelDownSampleAndTrace)
  compute_features   <- data_preparation.py:82-100 (jakteristics 0.5.1, verticality, search radius 0.6 m)
  propagate_to_original <- util/pipeline.py:441-452 (`propagate_preds_hash_full`: hash dictionary voxel -> original indices)

open3d and jakteristics are not part of the reference tree and not installable here: their published algorithms are
restated (csrc/tl_prepare.hip) and checked against an independent numpy/scipy oracle; parity with the two libraries
themselves is unpinned.  Deliberate differences, all on quantities the path does not depend on: voxels come out in
ascending (x, y, z) order (open3d: unordered_map order), and points with fewer than three neighbours inside the search
radius get NaN -> column mean (jakteristics diagonalises a rank-deficient covariance there)."""
import ctypes

import numpy as np
import torch

from .. import _hip

_I64x3 = ctypes.c_int64 * 3
_I64x2 = ctypes.c_int64 * 2


def _dev_f64(a):
    t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device="cuda", dtype=torch.float64).contiguous()


def _keys(xyz, cell, origin, round_input):
    """Packed cell keys (device i64[n]) of f64 points relative to `origin` (host float), plus the largest cell index per axis."""
    L = _hip.lib()
    n = len(xyz)
    src = torch.round(xyz * 100.0) / 100.0 if round_input else xyz
    base = [int(np.floor((float(v) - origin) / cell)) for v in src.amin(0).cpu()]
    top = [int(np.floor((float(v) - origin) / cell)) for v in src.amax(0).cpu()]
    keys = torch.empty(n, dtype=torch.int64, device=xyz.device); err = torch.empty(1, dtype=torch.int32, device=xyz.device)
    _hip.check(L.tl_cell_keys(_hip.ptr(xyz), n, float(cell), float(origin), _I64x3(*base), int(round_input), _hip.ptr(keys), _hip.ptr(err), _hip.stream()),
               "tl_cell_keys")
    if int(err.item()):
        raise ValueError("plot extent exceeds 2^21 cells along one axis")
    return keys, [t - b for t, b in zip(top, base)]


def voxelize(data, voxel_size):
    """data [N, 3 (+ other columns)] -> (down-sampled [M, 3 (+ other)] float64 like the reference's np.hstack, trace).
    Coordinates: mean of the 2-decimal-rounded points of each voxel (in input order, double), then float32 and rounded to
    2 decimals as `generate_tiles` does (util/pipeline.py:44-45); other columns: those of the first point of the voxel.
    `trace` = dict(first_idx i64[M], point2vox i64[N]) on the device (the reference returns open3d's per-voxel index lists)."""
    L = _hip.lib()
    arr = data if torch.is_tensor(data) else torch.from_numpy(np.ascontiguousarray(data))
    xyz = _dev_f64(arr[:, :3])
    n = len(xyz)
    bound = float((torch.round(xyz * 100.0) / 100.0).abs().max()) + 100.0             # data_preparation.py:68-69
    origin = -bound - 0.5 * voxel_size                                                # open3d shifts the grid by half a voxel
    keys, _ = _keys(xyz, voxel_size, origin, True)
    skeys, perm = torch.sort(keys, stable=True)
    out = torch.empty((n, 3), dtype=torch.float32, device=xyz.device)
    first = torch.empty(n, dtype=torch.int64, device=xyz.device); p2v = torch.empty(n, dtype=torch.int64, device=xyz.device)
    m = torch.empty(1, dtype=torch.int64, device=xyz.device)
    ws = torch.empty(int(L.tl_downsample_ws_words(n)), dtype=torch.int32, device=xyz.device)
    _hip.check(L.tl_downsample_reduce(_hip.ptr(xyz), _hip.ptr(skeys), _hip.ptr(perm), n, _hip.ptr(out), _hip.ptr(first), _hip.ptr(p2v), _hip.ptr(m),
                                      _hip.ptr(ws), _hip.stream()), "tl_downsample_reduce")
    M = int(m.item())
    pts = out[:M]; first = first[:M]
    if arr.shape[1] > 3:
        other = arr[:, 3:].to(xyz.device).index_select(0, first)
        pts = torch.cat([pts.double(), other.double()], 1)
    return pts, dict(first_idx=first, point2vox=p2v)


def compute_features(points, search_radius=0.6, feature_names=("verticality",)):
    """points [N,3] -> float32 [N,1] verticality, NaNs replaced by the mean of the finite values (replace_nanfeatures)."""
    assert list(feature_names) == ["verticality"], "only the verticality feature is on the path (util/pipeline.py:64)"
    L = _hip.lib()
    xyz = _dev_f64(points)
    n = len(xyz)
    origin = float(xyz.min()) - search_radius
    keys, ext = _keys(xyz, search_radius, origin, False)
    skeys, perm = torch.sort(keys)
    sx = xyz.index_select(0, perm).contiguous()
    v = torch.empty(n, dtype=torch.float32, device=xyz.device)
    _hip.check(L.tl_verticality(_hip.ptr(sx), _hip.ptr(skeys), n, float(search_radius), _I64x2(ext[0], ext[1]), _hip.ptr(v), _hip.stream()), "tl_verticality")
    out = torch.empty_like(v); out[perm] = v
    nan = torch.isnan(out)
    if bool(nan.any()):
        out[nan] = out[~nan].double().mean().float() if bool((~nan).any()) else float("nan")
    return out.reshape(-1, 1)


def propagate_to_original(preds_voxelized, trace):
    """Every original point takes the prediction of its voxel (util/pipeline.py:441-452 without the hash dictionary)."""
    p = preds_voxelized if torch.is_tensor(preds_voxelized) else torch.from_numpy(np.asarray(preds_voxelized))
    return p.to(trace["point2vox"].device).index_select(0, trace["point2vox"])
