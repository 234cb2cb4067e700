"""Seeded synthetic forest tiles (SURVEY.md §8d generator spec).

Produces the reference's tile npz schema (`points f32[N,3]`, `feat f32[N,1]`,
`instance_label i32[N]`, `center f64[3]` -- reference
tree_learn/util/data_preparation.py:476-480) so the tile loop, the parity
tests and bench.py all consume the same kind of input the reference's
TreeDataset (tree_learn/dataset/dataset.py:34-76) would load from disk.

Points are tile-centred float32 and de-duplicated to one point per global
voxel floor(p/v), mimicking data_preparation.py:60-79.
"""
import numpy as np

# BASELINE.json configs
CONFIGS = {
    "config1": dict(extent=10.0, voxel=0.2, n_trees=4, fill=0.10),
    "config2": dict(extent=40.0, voxel=0.1, n_trees=64, fill=0.10),
    "config5": dict(extent=40.0, voxel=0.05, n_trees=64, fill=0.12),
}


def make_tile(extent=40.0, voxel=0.1, n_trees=64, fill=0.10, seed=0):
    """Return dict(points f32[N,3], feat f32[N,1], instance_label i32[N], center f64[3])."""
    rng = np.random.default_rng(seed)
    E, v = float(extent), float(voxel)
    pts, lab, vert = [], [], []

    # ground: 1.5*(E/v)^2 points, z = 0.5 sin(x/7) + 0.3 cos(y/5) + N(0, 0.03)
    n_g = int(1.5 * (E / v) ** 2)
    xy = rng.uniform(-E / 2, E / 2, size=(n_g, 2))
    z = 0.5 * np.sin(xy[:, 0] / 7) + 0.3 * np.cos(xy[:, 1] / 5) + rng.normal(0, 0.03, n_g)
    pts.append(np.column_stack([xy, z])); lab.append(np.zeros(n_g, np.int32)); vert.append(rng.uniform(0, 1, n_g))

    # understory: 60*E^2 points, z in U(0, 1.5)
    n_u = int(60 * E * E)
    xy = rng.uniform(-E / 2, E / 2, size=(n_u, 2))
    pts.append(np.column_stack([xy, rng.uniform(0, 1.5, n_u)])); lab.append(np.zeros(n_u, np.int32)); vert.append(rng.uniform(0, 1, n_u))

    # trees: trunk cylinder surface + crown ellipsoid volume
    for t in range(n_trees):
        cx, cy = rng.uniform(-E / 2, E / 2, 2)
        h = rng.uniform(15, 30); r = rng.uniform(0.12, 0.35)
        n_t = int(2 * (2 * np.pi * r * h / (v * v)))
        th = rng.uniform(0, 2 * np.pi, n_t); tz = rng.uniform(0, h, n_t)
        pts.append(np.column_stack([cx + r * np.cos(th), cy + r * np.sin(th), tz]))
        lab.append(np.full(n_t, t + 1, np.int32)); vert.append(np.ones(n_t))
        cr = rng.uniform(2.5, 4.5); ch = h * rng.uniform(0.35, 0.55)
        vol = 4.0 / 3.0 * np.pi * cr * cr * (ch / 2)
        n_c = int(fill * vol / v ** 3)
        # uniform inside an ellipsoid: direction * u^(1/3)
        d = rng.normal(size=(n_c, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        rad = rng.uniform(0, 1, n_c) ** (1.0 / 3.0)
        d *= rad[:, None]
        pts.append(np.column_stack([cx + cr * d[:, 0], cy + cr * d[:, 1], (h - ch / 2) + (ch / 2) * d[:, 2]]))
        lab.append(np.full(n_c, t + 1, np.int32)); vert.append(rng.uniform(0, 1, n_c))

    p = np.concatenate(pts); l = np.concatenate(lab); f = np.concatenate(vert)
    keep = (np.abs(p[:, 0]) <= E / 2) & (np.abs(p[:, 1]) <= E / 2)
    p, l, f = p[keep], l[keep], f[keep]
    # one point per global voxel floor(p/v)  (data_preparation.py:60-79 analogue)
    c = np.floor(p / v).astype(np.int64)
    c -= c.min(axis=0)
    key = (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]
    _, first = np.unique(key, return_index=True)
    first.sort()
    p, l, f = p[first], l[first], f[first]
    return dict(points=p.astype(np.float32), feat=f.astype(np.float32)[:, None],
                instance_label=l.astype(np.int32), center=np.zeros(3, np.float64))


def make_batch(tiles, inner_square_edge_length=8.0):
    """Collate tiles into the reference's batch dict (dataset.py:167-226), CPU numpy->torch.

    Labels/masks follow dataset.py:41-66 in spirit (offset labels are a cheap
    stand-in: tree base = mean of lowest 0.5 m), enough for loss plumbing.
    """
    import torch
    xyzs, feats, bids, inst, sem, offl, cen, m_in, m_off, m_sem = [], [], [], [], [], [], [], [], [], []
    for b, t in enumerate(tiles):
        xyz = t["points"]; il = t["instance_label"].astype(np.int64)
        sl = np.where(il == 0, 1, 0).astype(np.int64)
        pos = np.ones_like(xyz)
        valid = np.zeros(len(xyz), bool)
        for i in np.unique(il):
            if i == 0:
                continue
            idx = np.where(il == i)[0]
            tp = xyz[idx]
            lo = tp[tp[:, 2] <= tp[:, 2].min() + 0.5]
            pos[idx] = lo.mean(axis=0); valid[idx] = True
        mi = np.linalg.norm(xyz[:, :2], ord=np.inf, axis=1) <= inner_square_edge_length / 2
        xyzs.append(xyz); feats.append(t["feat"]); bids.append(np.full(len(xyz), b, np.int64))
        inst.append(il); sem.append(sl); offl.append((pos - xyz).astype(np.float32))
        cen.append(np.ones_like(xyz) * t["center"].astype(np.float32))
        m_in.append(mi); m_off.append(mi & (sl != 1) & valid); m_sem.append(mi)
    cat = lambda xs: torch.from_numpy(np.concatenate(xs, 0))
    return {
        "coords": cat(xyzs).float(), "input_feats": cat(feats).float(), "batch_ids": cat(bids).long(),
        "semantic_labels": cat(sem).long(), "instance_labels": cat(inst).long(),
        "masks_inner": cat(m_in).bool(), "masks_off": cat(m_off).bool(), "masks_sem": cat(m_sem).bool(),
        "offset_labels": cat(offl).float(), "batch_size": len(tiles), "centers": cat(cen).float(),
    }
