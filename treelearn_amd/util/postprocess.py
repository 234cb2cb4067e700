"""SURVEY.md §8f "next" rows on the GPU: `ensemble` (reference tree_learn/util/pipeline.py:113-141) and
`assign_remaining_points_nearest_neighbor` (:287-296)."""
import ctypes

import numpy as np
import torch

from .. import _hip


def ensemble(coords, semantic_scores, semantic_labels, offset_predictions, offset_labels, instance_labels, feats, input_feats,
             device="cuda", return_device=False):
    """Mean of duplicate predictions keyed by coords rounded to 0.01 m, output sorted by (x, y, z).

    Reference: pandas `df.round({'x':2,'y':2,'z':2}).groupby(['x','y','z']).mean()`.  Here: the float32 rounding
    np.round(x, 2) = rint(x * 100) / 100 gives integer keys; one stable device sort, then `tl_group_mean`: one thread per
    group adds its members in input order in fp64 (as the groupby kernels do) over all 40-odd columns at once; integer label
    columns are averaged as floats then truncated, like the reference (:130,135,138).
    """
    L = _hip.lib()
    T = lambda a: a.to(device) if torch.is_tensor(a) else torch.as_tensor(np.ascontiguousarray(a)).to(device)   # noqa: E731
    c = T(coords).float()
    q = torch.round(c * 100.0)                                    # float32 multiply + rint == np.round(., 2) numerator
    qi = q.to(torch.int64)
    r = qi - qi.min(dim=0).values
    span = r.max(dim=0).values + 1
    key = (r[:, 0] * span[1] + r[:, 1]) * span[2] + r[:, 2]       # lexicographic (x, y, z)
    skey, perm = torch.sort(key, stable=True)
    cols = [semantic_scores, semantic_labels, offset_predictions, offset_labels, instance_labels, feats, input_feats]
    mats = [T(a).reshape(len(key), -1).float() for a in cols]
    widths = [m.shape[1] for m in mats]
    src = torch.cat(mats, 1).contiguous()                         # one [N, 2+1+3+3+1+32+F] matrix -> one fused kernel
    n, C = src.shape
    mean = torch.empty((n, C), dtype=torch.float64, device=src.device)
    first = torch.empty(n, dtype=torch.int64, device=src.device); m = torch.empty(1, dtype=torch.int64, device=src.device)
    ws = torch.empty(int(L.tl_downsample_ws_words(n)), dtype=torch.int32, device=src.device)
    _hip.check(L.tl_group_mean(_hip.ptr(src), n, C, _hip.ptr(skey), _hip.ptr(perm), _hip.ptr(mean), _hip.ptr(first), _hip.ptr(m), _hip.ptr(ws),
                               _hip.stream()), "tl_group_mean")
    M = int(m.item())
    mean = mean[:M]; first = first[:M]
    parts = torch.split(mean, widths, dim=1)
    out_coords = (q[first] / 100.0).float()                       # the rounded coordinates themselves
    if return_device:            # same eight results as device tensors (the 32 backbone columns are the bulk of the D2H otherwise)
        f32 = lambda t: t.float()                                                            # noqa: E731
        i64 = lambda t: t[:, 0].to(torch.int64)                                              # noqa: E731
    else:
        f32 = lambda t: t.float().cpu().numpy()                                              # noqa: E731
        i64 = lambda t: t[:, 0].cpu().numpy().astype('int64').flatten()                      # noqa: E731
    one = lambda t, a: t[:, 0] if a.ndim == 1 else t                                         # noqa: E731
    return (out_coords if return_device else out_coords.cpu().numpy(),
            f32(one(parts[0], semantic_scores)), i64(parts[1]), f32(one(parts[2], offset_predictions)), f32(one(parts[3], offset_labels)),
            i64(parts[4]), f32(one(parts[5], feats)), f32(one(parts[6], input_feats)))


GRID_KNN_MIN_REF = 4096          # below this the brute-force kernel is as fast


def knn_vote(ref, lab, qry, k, force=None):
    """Majority label of the k nearest reference points of every query (device tensors: ref f32[nr,3], lab i64[nr], qry f32[nq,3]).
    Large reference sets go through the cell grid (tl_knn_vote_grid: exact, bit-identical to the brute-force tl_knn_vote);
    `force` = "grid" / "brute" pins the path (tests)."""
    L = _hip.lib()
    nr, nq = ref.shape[0], qry.shape[0]
    if k < 1 or k > nr:                                                # sklearn's rule (and message) for the classifier / NearestNeighbors the reference uses
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, n_samples_fit = {nr}" if k >= 1 else f"Expected n_neighbors > 0. Got {k}")
    out = torch.empty(nq, dtype=torch.int64, device=ref.device)
    if nq == 0:
        return out
    use_grid = force == "grid" or (force is None and nr >= GRID_KNN_MIN_REF)
    if not use_grid:
        _hip.check(L.tl_knn_vote(_hip.ptr(ref), _hip.ptr(lab), nr, _hip.ptr(qry), nq, int(k), _hip.ptr(out), _hip.stream()), "tl_knn_vote")
        return out
    lo = ref.min(dim=0).values
    ext = (ref.max(dim=0).values - lo).clamp_min(1e-3)
    lo_h, ext_h = lo.cpu().numpy().astype(np.float32), ext.cpu().numpy().astype(np.float64)
    # cell edge: about 8 reference points per cell of the bounding box, at least 5 cm, at most 2^20 cells per axis
    h = np.float32(max(0.05, float((ext_h.prod() * 8.0 / nr) ** (1.0 / 3.0)), float(ext_h.max()) / (1 << 20)))
    inv_h = np.float32(1.0) / h                                        # the kernel computes 1.0f / h the same way
    cell = ((ref - lo) * float(inv_h)).floor().to(torch.int64)
    dims = (cell.max(dim=0).values + 1).cpu().numpy().astype(np.int64)
    key = (cell[:, 0] * int(dims[1]) + cell[:, 1]) * int(dims[2]) + cell[:, 2]
    skey, order = torch.sort(key, stable=True)
    ukeys, counts = torch.unique_consecutive(skey, return_counts=True)
    starts = torch.zeros(len(ukeys) + 1, dtype=torch.int64, device=ref.device)
    starts[1:] = torch.cumsum(counts, 0)
    rs, ls = ref.index_select(0, order).contiguous(), lab.index_select(0, order).contiguous()
    _hip.check(L.tl_knn_vote_grid(_hip.ptr(rs), _hip.ptr(ls), _hip.ptr(order), nr, _hip.ptr(ukeys), _hip.ptr(starts), len(ukeys),
                                  (ctypes.c_float * 3)(*[float(v) for v in lo_h]), float(h), _hip.dims3(dims), _hip.ptr(qry), nq, int(k),
                                  _hip.ptr(out), _hip.stream()), "tl_knn_vote_grid")
    return out


def assign_remaining_points_nearest_neighbor(coords, predictions, remaining_points_idx, n_neighbors=5, device="cuda"):
    """Unassigned points (label == remaining_points_idx) take the majority label of their k nearest assigned points
    (exact brute-force k-NN on the GPU, tl_knn_vote)."""
    L = _hip.lib()
    predictions = np.copy(predictions)
    assert len(coords) == len(predictions)
    qi = np.argwhere(predictions == remaining_points_idx).reshape(-1)
    ri = np.argwhere(predictions != remaining_points_idx).reshape(-1)
    if len(qi) == 0:
        return predictions.astype(np.int64)
    ref = torch.from_numpy(np.ascontiguousarray(coords[ri], dtype=np.float32)).to(device)
    lab = torch.from_numpy(np.ascontiguousarray(predictions[ri]).astype(np.int64)).to(device)
    qry = torch.from_numpy(np.ascontiguousarray(coords[qi], dtype=np.float32)).to(device)
    predictions[qi] = knn_vote(ref, lab, qry, n_neighbors).cpu().numpy()
    return predictions.astype(np.int64)


def assign_remaining_points_nearest_neighbor_device(coords, predictions, remaining_points_idx, n_neighbors=5):
    """The same k-NN fill for DEVICE tensors (coords f32 [n, 3], predictions i64 [n]); returns an int64 device tensor."""
    predictions = predictions.clone()
    q = predictions == remaining_points_idx
    qi, ri = torch.nonzero(q).squeeze(1), torch.nonzero(~q).squeeze(1)
    if qi.numel() == 0:
        return predictions
    c = coords.float().contiguous()
    predictions[qi] = knn_vote(c.index_select(0, ri).contiguous(), predictions.index_select(0, ri).contiguous(), c.index_select(0, qi).contiguous(), n_neighbors)
    return predictions


def propagate_preds(source_coords, source_preds, target_coords, n_neighbors, n_jobs=1, device="cuda"):
    """Every target point takes the most frequent label among its k nearest source points, the smallest label on ties
    (reference util/pipeline.py:300-331: sklearn NearestNeighbors + np.bincount(...).argmax()); exact k-NN on the GPU."""
    L = _hip.lib()
    ref = torch.from_numpy(np.ascontiguousarray(source_coords, dtype=np.float32)).to(device)
    lab = torch.from_numpy(np.ascontiguousarray(source_preds).astype(np.int64)).to(device)
    qry = torch.from_numpy(np.ascontiguousarray(target_coords, dtype=np.float32)).to(device)
    return knn_vote(ref, lab, qry, n_neighbors).cpu().numpy()
