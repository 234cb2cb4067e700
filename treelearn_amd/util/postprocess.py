"""SURVEY.md §8f "next" rows on the GPU: `ensemble` (reference tree_learn/util/pipeline.py:113-141) and
`assign_remaining_points_nearest_neighbor` (:287-296)."""
import numpy as np
import torch

from .. import _hip


def ensemble(coords, semantic_scores, semantic_labels, offset_predictions, offset_labels, instance_labels, feats, input_feats,
             device="cuda"):
    """Mean of duplicate predictions keyed by coords rounded to 0.01 m, output sorted by (x, y, z).

    Reference: pandas `df.round({'x':2,'y':2,'z':2}).groupby(['x','y','z']).mean()`.  Here: the float32 rounding
    np.round(x, 2) = rint(x * 100) / 100 gives integer keys; one device sort + segmented mean (fp64 accumulation, as the
    groupby kernels do); integer label columns are averaged as floats then truncated, like the reference (:130,135,138).
    """
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(device)                       # noqa: E731
    c = T(coords).float()
    q = torch.round(c * 100.0)                                    # float32 multiply + rint == np.round(., 2) numerator
    qi = q.to(torch.int64)
    off = qi.min(dim=0).values
    r = qi - off
    span = r.max(dim=0).values + 1
    key = (r[:, 0] * span[1] + r[:, 1]) * span[2] + r[:, 2]       # lexicographic (x, y, z)
    ukey, inv = torch.unique(key, sorted=True, return_inverse=True)
    n = ukey.shape[0]
    cnt = torch.zeros(n, dtype=torch.float64, device=c.device).index_add_(0, inv, torch.ones_like(key, dtype=torch.float64))

    def gmean(a):
        a = T(a)
        two_d = a.dim() == 2
        a2 = a.reshape(a.shape[0], -1).double()
        s = torch.zeros((n, a2.shape[1]), dtype=torch.float64, device=c.device).index_add_(0, inv, a2)
        m = s / cnt[:, None]
        return m if two_d else m[:, 0]

    first = torch.zeros(n, dtype=torch.int64, device=c.device).scatter_reduce_(0, inv, torch.arange(len(key), device=c.device), "amin", include_self=False)
    out_coords = (q[first] / 100.0).float()                       # the rounded coordinates themselves
    res = (out_coords.cpu().numpy(),
           gmean(semantic_scores).float().cpu().numpy(),
           gmean(semantic_labels).cpu().numpy().astype('int64').flatten(),
           gmean(offset_predictions).float().cpu().numpy(),
           gmean(offset_labels).float().cpu().numpy(),
           gmean(instance_labels).cpu().numpy().astype('int64').flatten(),
           gmean(feats).float().cpu().numpy(),
           gmean(input_feats).float().cpu().numpy())
    return res


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
    out = torch.empty(len(qi), dtype=torch.int64, device=ref.device)
    _hip.check(L.tl_knn_vote(_hip.ptr(ref), _hip.ptr(lab), len(ri), _hip.ptr(qry), len(qi), int(n_neighbors), _hip.ptr(out), _hip.stream()), "tl_knn_vote")
    predictions[qi] = out.cpu().numpy()
    return predictions.astype(np.int64)
