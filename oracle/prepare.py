"""Oracle (test infrastructure only -- see oracle/__init__.py) for SURVEY.md 8f #4: the global voxel down-sample and the
verticality feature.

PARITY UNPINNED: the reference delegates both to third-party libraries that are neither in /root/reference nor in this
image -- open3d==0.17.0 `PointCloud.voxel_down_sample_and_trace` (tree_learn/util/data_preparation.py:60-79) and
jakteristics==0.5.1 `compute_features` (data_preparation.py:82-88), pinned in setup/requirements.txt:9-10 -- and has no
tests or fixtures for them.  Restated from their published algorithms:
  * open3d: voxel index = floor((p - (min_bound - voxel/2)) / voxel); a voxel's point = the running double sum of its
    points in input order divided by their number; the trace lists the original indices per voxel (first = smallest).
    The OUTPUT ORDER of open3d is that of a std::unordered_map and is not reproduced: ascending (x, y, z) voxel index here.
  * jakteristics: neighbours = all points within the search radius (cKDTree.query_ball_point, inclusive, self included);
    sample covariance (n - 1); eigenvectors by descending eigenvalue; verticality = 1 - |z of the last eigenvector|.
The reference's own code around them (rounding to 2 decimals, float32 casts, NaN replacement) is followed literally:
data_preparation.py:62,68-79,91-100; util/pipeline.py:44-45."""
import numpy as np


def voxelize(data, voxel_size):
    data = np.asarray(data, np.float64)
    pts = np.round(data[:, :3], 2)
    bound = np.max(np.abs(pts)) + 100
    vmin = -bound - 0.5 * voxel_size
    vox = np.floor((pts - vmin) / voxel_size).astype(np.int64)
    order = np.lexsort((np.arange(len(pts)), vox[:, 2], vox[:, 1], vox[:, 0]))
    sv = vox[order]
    heads = np.flatnonzero(np.r_[True, np.any(sv[1:] != sv[:-1], axis=1)])
    ends = np.r_[heads[1:], len(pts)]
    out = np.empty((len(heads), 3)); first = np.empty(len(heads), np.int64); p2v = np.empty(len(pts), np.int64)
    for m, (a, b) in enumerate(zip(heads, ends)):
        acc = np.zeros(3)
        for o in order[a:b]:                 # input order, left-to-right double sum
            acc += pts[o]
        out[m] = acc / (b - a); first[m] = order[a]; p2v[order[a:b]] = m
    out = np.round(out.astype(np.float32), 2)
    if data.shape[1] > 3:
        out = np.hstack([out.astype(np.float64), data[first, 3:]])
    return out, first, p2v


def verticality(points, search_radius=0.6):
    from scipy.spatial import cKDTree
    pts = np.asarray(points, np.float64)
    tree = cKDTree(pts)
    out = np.full(len(pts), np.nan, np.float32)
    gap = np.zeros(len(pts))
    for i, nb in enumerate(tree.query_ball_point(pts, search_radius)):
        if len(nb) < 3:
            continue
        w, v = np.linalg.eigh(np.cov(pts[nb].T))
        out[i] = 1.0 - abs(v[2, 0])
        gap[i] = (w[1] - w[0]) / max(w[2], 1e-300)          # relative gap of the two smallest eigenvalues (conditioning of the normal)
    return out, gap


def replace_nan(features):
    f = np.array(features, np.float32)
    nan = np.isnan(f)
    if nan.any():
        f[nan] = np.nanmean(f)
    return f
