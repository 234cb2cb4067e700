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
    "config5": dict(extent=40.0, voxel=0.05, n_trees=64, fill=0.12),          # 15.2 M points (the parity tests' stress tile)
    "config5_20m": dict(extent=40.0, voxel=0.05, n_trees=64, fill=0.16),      # BASELINE config 5's "~20 M active voxels" (fill 0.16)
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


def tile_variant(tile, sym):
    """One of the 8 symmetries of the square applied to a tile (sym & 1: swap x/y, sym & 2: mirror x, sym & 4: mirror y).
    Gives 8 geometrically distinct tiles per generated tile at memcpy cost: the 64 tiles of BASELINE config 4 are the
    8 symmetries of seeds 0..7 (`plot_tiles`)."""
    p = tile["points"].copy()
    if sym & 1:
        p[:, [0, 1]] = p[:, [1, 0]]
    if sym & 2:
        p[:, 0] = -p[:, 0]
    if sym & 4:
        p[:, 1] = -p[:, 1]
    out = dict(tile); out["points"] = np.ascontiguousarray(p)
    return out


def plot_tile_ids(n_tiles=64):
    """(seed, symmetry) of tile i of the config-4 plot: tile i = symmetry i // 8 of generator seed i % 8 (so a round-robin
    shard over 2 / 4 / 8 ranks needs only 4 / 2 / 1 generated seeds per rank)."""
    return [(i % 8, i // 8) for i in range(n_tiles)]


def plot_tiles(indices, cfg=None, cache=None):
    """The config-4 tiles with the given indices (only these are generated -- a rank builds only its own tiles)."""
    cfg = cfg or CONFIGS["config2"]
    cache = {} if cache is None else cache
    out = []
    for i in indices:
        seed, sym = i % 8, i // 8
        if seed not in cache:
            cache[seed] = make_tile(**cfg, seed=seed)
        out.append(tile_variant(cache[seed], sym))
    return out


PLOT4 = dict(tiles_per_side=8, inner_edge=8.0, outer_edge=16.0, stride=0.5)     # BASELINE config 4: 64 overlapping 40 x 40 m tiles of ONE plot


def plot_squares(tiles_per_side=8, inner_edge=8.0, outer_edge=16.0, stride=0.5):
    """Inner and outer squares (x0, x1, y0, y1), float64 [T, 4], of the config-4 plot: tiles_per_side^2 inner squares of `inner_edge` laid
    out every stride * inner_edge (the reference's generate_tiles produces its inference tiles the same way: overlapping inner squares,
    each with its context of outer_edge on every side -- tree_learn/util/data_preparation.py:362-389, tools/pipeline/pipeline.py:52-70),
    row-major from the top-left corner, the plot centred on the origin.  Neighbouring 40 m tiles overlap by 36 m: every point of the plot's
    interior is predicted by up to four tiles' inner squares, which is what `ensemble` then averages."""
    step = stride * inner_edge
    span = (tiles_per_side - 1) * step + inner_edge
    x0 = -span / 2 + step * np.arange(tiles_per_side)
    y1 = span / 2 - step * np.arange(tiles_per_side)
    inner = np.stack([np.tile(x0, tiles_per_side), np.tile(x0 + inner_edge, tiles_per_side),
                      np.repeat(y1 - inner_edge, tiles_per_side), np.repeat(y1, tiles_per_side)], axis=1).astype(np.float64)
    outer = inner + np.array([-outer_edge, outer_edge, -outer_edge, outer_edge])
    return inner, outer


def make_plot(tiles_per_side=8, inner_edge=8.0, outer_edge=16.0, stride=0.5, voxel=0.1, fill=0.10, seed=0):
    """The ONE synthetic plot the config-4 tiles are cropped from: the config-2 generator over the union of all outer squares (68 x 68 m for
    the default 8 x 8 tiles), with the tree density of config 2 (64 trees per 40 x 40 m)."""
    inner, outer = plot_squares(tiles_per_side, inner_edge, outer_edge, stride)
    extent = float(outer[:, 1].max() - outer[:, 0].min())
    return make_tile(extent=extent, voxel=voxel, n_trees=int(round(64 * (extent / 40.0) ** 2)), fill=fill, seed=seed)


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
        # tree base per instance = mean of its points within 0.5 m of its lowest point (one grouped pass, float64 sums)
        uniq, inv = np.unique(il, return_inverse=True)
        zmin = np.full(len(uniq), np.inf, np.float32)
        np.minimum.at(zmin, inv, xyz[:, 2])
        low = xyz[:, 2] <= zmin[inv] + np.float32(0.5)
        cnt = np.bincount(inv[low], minlength=len(uniq)).astype(np.float64)
        base = np.stack([np.bincount(inv[low], weights=xyz[low, d].astype(np.float64), minlength=len(uniq)) / np.maximum(cnt, 1) for d in range(3)], 1)
        valid = (il != 0)
        pos = np.where(valid[:, None], base[inv].astype(np.float32), np.float32(1.0))
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


# ---------------------------------------------------------------- synthetic weights (random init of the reference architecture)
def state_dict_manifest(channels=32, num_blocks=7, dim_coord=3, dim_feat=1, kernel_size=3):
    """(key, shape) list in the reference's registration order."""
    m = []
    def bn(p, c):
        m.extend([(p + ".weight", (c,)), (p + ".bias", (c,)), (p + ".running_mean", (c,)),
                  (p + ".running_var", (c,)), (p + ".num_batches_tracked", ())])
    k = kernel_size
    m.append(("input_conv.0.weight", (channels, k, k, k, dim_coord + dim_feat)))
    planes = [channels * (i + 1) for i in range(num_blocks)]
    def resblock(p, cin, cout):
        if cin != cout:
            m.append((p + ".i_branch.0.weight", (cout, 1, 1, 1, cin)))
        bn(p + ".conv_branch.0", cin)
        m.append((p + ".conv_branch.2.weight", (cout, k, k, k, cin)))
        bn(p + ".conv_branch.3", cout)
        m.append((p + ".conv_branch.5.weight", (cout, k, k, k, cout)))
    def ub(p, pl):
        for i in range(2):
            resblock(f"{p}.blocks.block{i}", pl[0], pl[0])
        if len(pl) > 1:
            bn(p + ".conv.0", pl[0]); m.append((p + ".conv.2.weight", (pl[1], 2, 2, 2, pl[0])))
            ub(p + ".u", pl[1:])
            bn(p + ".deconv.0", pl[1]); m.append((p + ".deconv.2.weight", (pl[0], 2, 2, 2, pl[1])))
            for i in range(2):
                resblock(f"{p}.blocks_tail.block{i}", pl[0] * (2 - i), pl[0])
    ub("unet", planes)
    bn("output_layer.0", channels)
    for name, co in (("semantic_linear", 2), ("offset_linear", 3)):
        m.append((name + ".0.weight", (channels, channels))); m.append((name + ".0.bias", (channels,)))
        bn(name + ".1", channels)
        m.append((name + ".3.weight", (co, channels))); m.append((name + ".3.bias", (co,)))
    return m


def random_state_dict(seed, **cfg):
    """Deterministic non-trivial weights/BN statistics, generated key by key from a numpy
    Generator so tests and the golden script agree without storing 30 M parameters."""
    import torch
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape in state_dict_manifest(**cfg):
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.tensor(7, dtype=torch.long); continue
        if key.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape)
        elif key.endswith("running_mean"):
            v = rng.normal(0, 0.2, shape)
        elif key.endswith(".bias"):
            v = rng.normal(0, 0.1, shape)
        elif len(shape) == 1:                               # BN weight
            v = rng.uniform(0.7, 1.3, shape)
        elif len(shape) == 5:                               # conv: keep activations O(1)
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            v = rng.normal(0, (2.0 / fan_in) ** 0.5 * 1.5, shape)
        else:                                               # linear
            v = rng.normal(0, (1.0 / shape[1]) ** 0.5, shape)
        sd[key] = torch.from_numpy(np.asarray(v, np.float32))
    return sd
