"""Oracle: voxel hashing + rulebook construction (numpy, integer exact).

Canonical formats: SURVEY.md Appendix F.
  key   = b<<48 | x<<32 | y<<16 | z          (uint64 held in int64)
  order = ascending key; coords i32[M,4] = (b,x,y,z); v2p = rank of the point's key
Follows reference tree_learn/model/tree_learn.py:129-167 (voxelize) and the
spconv PointToVoxel / SubMConv3d / SparseConv3d index semantics restated in
SURVEY.md §8b + Appendix B (spconv itself is not available here).
"""
import numpy as np

MAX_POINTS_DEFAULT = 3


def pack_key(c):
    c = c.astype(np.int64)
    return (c[:, 0] << 48) | (c[:, 1] << 32) | (c[:, 2] << 16) | c[:, 3]


def point_to_voxel_coords(coords, batch_ids, batch_size, voxel_size, epsilon=1.0):
    """Integer voxel coordinate of every point, per batch element.

    tree_learn.py:133-143: range = [min, max+epsilon] over the element's points,
    PointToVoxel computes c = floor((p - min) / vsize) in fp32 and accepts it iff
    0 <= c < grid, grid = round((max+eps - min)/vsize) (spconv PointToVoxel).
    Returns c i32[N,4] = (b,x,y,z) and valid bool[N].
    """
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    batch_ids = np.asarray(batch_ids).astype(np.int64)
    vs = np.float32(voxel_size)
    out = np.zeros((len(coords), 4), np.int32)
    valid = np.zeros(len(coords), bool)
    for b in range(batch_size):
        sel = np.where(batch_ids == b)[0]
        if len(sel) == 0:
            continue
        p = coords[sel]
        mn = p.min(axis=0)                                   # fp32
        mx = p.max(axis=0) + np.float32(epsilon)             # fp32 add, tree_learn.py:135
        # python floats of the fp32 values go into spconv, which holds them as float
        grid = np.round((mx.astype(np.float64) - mn.astype(np.float64)) / float(vs)).astype(np.int64)
        c = np.floor((p - mn) / vs).astype(np.int64)         # fp32 sub, fp32 div, floor
        ok = np.all((c >= 0) & (c < grid), axis=1)
        out[sel, 0] = b
        out[sel, 1:] = c
        valid[sel] = ok
    return out, valid


def voxelize(coords, feats, batch_ids, batch_size, voxel_size, use_coords=False, use_feats=False,
             max_num_points_per_voxel=MAX_POINTS_DEFAULT, epsilon=1.0):
    """Restates tree_learn.py:129-167.

    Returns voxel_feats f32[M,C] in (feat.., x,y,z) column order, voxel_coords i32[M,4],
    v2p i64[N], spatial_shape i64[3] (= max+1 over all voxels, tree_learn.py:165).
    Voxel order = ascending key (the reference's GPU order is hash-slot order).
    Features: mean over the first <= P points of the voxel in input order, all-zero
    point rows treated as empty slots (tree_learn.py:149-151); ones when disabled.
    """
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    feats = np.ascontiguousarray(feats, dtype=np.float32).reshape(len(coords), -1)
    c, valid = point_to_voxel_coords(coords, batch_ids, batch_size, voxel_size, epsilon)
    assert valid.all(), "point outside voxel range (reference asserts at tree_learn.py:144)"
    key = pack_key(c)
    ukeys, first, v2p = np.unique(key, return_index=True, return_inverse=True)
    vcoords = c[first].astype(np.int32)
    M = len(ukeys)
    pf = np.concatenate([coords, feats], axis=1)             # (x,y,z,feat) tree_learn.py:85
    C = pf.shape[1]
    if use_coords or use_feats:
        order = np.argsort(v2p, kind="stable")               # points grouped by voxel, input order kept
        sv = v2p[order]
        start = np.searchsorted(sv, np.arange(M))
        rank = np.arange(len(sv)) - start[sv]
        keep = rank < max_num_points_per_voxel
        rows = pf[order][keep]; vox = sv[keep]
        nonzero = ~(rows == 0).all(axis=1)                   # zero rows count as padding
        s = np.zeros((M, C), np.float64); n = np.zeros(M, np.float64)
        np.add.at(s, vox[nonzero], rows[nonzero].astype(np.float64))
        np.add.at(n, vox[nonzero], 1.0)
        with np.errstate(invalid="ignore", divide="ignore"):
            mean = (s / n[:, None]).astype(np.float32)       # NaN when all rows zero (nanmean of all-NaN)
    else:
        mean = np.ones((M, C), np.float32)
    if not use_coords:
        mean[:, :3] = 1.0
    if not use_feats:
        mean[:, 3:] = 1.0
    vfeats = np.concatenate([mean[:, 3:], mean[:, :3]], axis=1).astype(np.float32)
    spatial_shape = vcoords[:, 1:].max(axis=0).astype(np.int64) + 1
    return vfeats, vcoords, v2p.astype(np.int64), spatial_shape


def _lookup(sorted_keys, q):
    pos = np.searchsorted(sorted_keys, q)
    pos_c = np.minimum(pos, len(sorted_keys) - 1)
    hit = sorted_keys[pos_c] == q
    return np.where(hit, pos_c, -1).astype(np.int32)


def rulebook_subm(vcoords, ksize=3):
    """SubMConv3d(k, padding=k//2) neighbour table i32[M, k^3], tap = (a*k + b)*k + c with
    (a,b,c) the kernel index along (x,y,z); in_pos = out_pos - pad + tap (cross-correlation,
    Appendix B).  Entry = input row or -1.  Neighbours never cross batch elements or leave
    the non-negative lattice.  (spatial upper bound: a voxel beyond it cannot be active.)"""
    keys = pack_key(vcoords)
    assert np.all(np.diff(keys) > 0), "coords must be sorted by key and unique"
    pad = ksize // 2
    M = len(vcoords)
    nbr = np.full((M, ksize ** 3), -1, np.int32)
    c = vcoords.astype(np.int64)
    for a in range(ksize):
        for b in range(ksize):
            for d in range(ksize):
                q = c.copy()
                q[:, 1] += a - pad; q[:, 2] += b - pad; q[:, 3] += d - pad
                ok = (q[:, 1:] >= 0).all(axis=1) & (q[:, 1:] < 65536).all(axis=1)
                r = _lookup(keys, pack_key(np.where(ok[:, None], q, 0)))
                nbr[:, (a * ksize + b) * ksize + d] = np.where(ok, r, -1)
    return nbr


def rulebook_down(vcoords, in_shape):
    """SparseConv3d(k=2, s=2, p=0): coarse coords (ascending key), parent i32[M] (fine->coarse
    row, -1 if the fine voxel falls beyond out_shape), child i32[M', 8] with tap
    k = (x&1)*4 + (y&1)*2 + (z&1), out_shape = in_shape // 2 (Appendix B/F).
    Raises ValueError("... reach zero!!!") when an output dim collapses to 0, the condition
    under which spconv aborts and the reference tile loop skips the tile
    (tree_learn/util/pipeline.py:91-97)."""
    in_shape = np.asarray(in_shape, np.int64)
    out_shape = in_shape // 2
    if (out_shape <= 0).any():
        raise ValueError(f"sparse conv output spatial shape {out_shape.tolist()} reach zero!!! "
                         f"input shape {in_shape.tolist()}")
    c = vcoords.astype(np.int64)
    q = c.copy(); q[:, 1:] >>= 1
    inb = (q[:, 1:] < out_shape[None, :]).all(axis=1)
    qk = pack_key(q)
    ck = np.unique(qk[inb])
    parent = np.where(inb, _lookup(ck, qk), -1).astype(np.int32)
    Mc = len(ck)
    ccoords = np.stack([(ck >> 48) & 0xFFFF, (ck >> 32) & 0xFFFF, (ck >> 16) & 0xFFFF, ck & 0xFFFF], axis=1).astype(np.int32)
    tap = ((c[:, 1] & 1) * 4 + (c[:, 2] & 1) * 2 + (c[:, 3] & 1)).astype(np.int64)
    child = np.full((Mc, 8), -1, np.int32)
    rows = np.where(inb)[0]
    child[parent[rows], tap[rows]] = rows.astype(np.int32)
    return ccoords, parent, child, out_shape
