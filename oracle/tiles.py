"""Oracle (test infrastructure only -- see oracle/__init__.py): numpy restatement of the reference's inference tiling,
SURVEY.md 8f #3.  Follows
  * SampleGenerator.tile_generate_and_save, tree_learn/util/data_preparation.py:333-494 (plot_corners=None, no denoising:
    configs/_modular/sample_generation.yaml:9-13 leaves the SOR / radius filters off and the pipeline never passes corners,
    tree_learn/util/pipeline.py:74-75),
  * TreeDataset.__getitem__ (test mode) / getOffset / get_mask_inner / collate_fn with batch size 1,
    tree_learn/dataset/dataset.py:34-140,167-226.
Pinned by tests/golden/g11_tiles.npz (made by running those reference functions; tests/golden/make_golden.py).

The dtype walk matters for bit-exactness and is restated literally:
  plot points/labels/features are float32; the outer-square test runs on the GPU in torch where the float64 bounds are
  0-dim tensors and therefore compared in float32; the inner-square test runs in numpy against np.float64 scalars, i.e. in
  float64; chunk centres are computed in float32 (the kept inner extensions are cast to float32 first) and subtracted in
  float64; the result is stored as float32.
"""
import numpy as np

IGNORE_RAW, NON_TREE_RAW = -1, 0          # dataset.py:7-8
NON_TREE_DS, TREE_DS = 1, 0               # dataset.py:9-10


def tile_grid(x_range, y_range, inner_edge, outer_edge, stride):
    """Inner/outer square extensions [T,4] = (xmin, xmax, ymin, ymax), float64, row-major from the top-left tile
    (data_preparation.py:362-389)."""
    xmin = np.round(x_range[0] - 1.5 * outer_edge, 2); xmax = np.round(x_range[1] + 1.5 * outer_edge, 2)
    ymin = np.round(y_range[0] - 1.5 * outer_edge, 2); ymax = np.round(y_range[1] + 1.5 * outer_edge, 2)
    ncols = int(np.round((xmax - xmin - 2 * outer_edge) / inner_edge))
    ex = np.round((xmax - xmin - 2 * outer_edge) / ncols, 5)
    ncols = int((ncols - 1) / stride + 1)
    nrows = int(np.round((ymax - ymin - 2 * outer_edge) / inner_edge))
    ey = np.round((ymax - ymin - 2 * outer_edge) / nrows, 5)
    nrows = int((nrows - 1) / stride + 1)
    inner = np.empty((nrows * ncols, 4))
    for i in range(nrows):
        for j in range(ncols):
            inner[i * ncols + j] = (xmin + outer_edge + stride * j * ex, xmin + outer_edge + (stride * j + 1) * ex,
                                    ymax - outer_edge - (stride * i + 1) * ey, ymax - outer_edge - stride * i * ey)
    inner = np.round(inner, 5)
    outer = inner + np.array([-outer_edge, outer_edge, -outer_edge, outer_edge]).reshape(1, 4)
    return inner, outer


def offset_labels(xyz, instance_label, semantic_label):
    """dataset.py:111-140: per tree instance, base = float32 mean of its points not higher than 0.5 m above a low z
    (np.partition(z, 10)[3] when it has more than 11 points, else the minimum); label = base - xyz."""
    position = np.ones_like(xyz, dtype=np.float32)
    valid = np.zeros(len(xyz), dtype=bool)
    for inst in np.unique(instance_label):
        sel = np.where(instance_label == inst)[0]
        if semantic_label[sel[0]] == NON_TREE_DS:
            continue
        pts = xyz[sel]
        z = pts[:, 2]
        low = np.partition(z, 10)[3] if len(z) > 11 else z.min()
        base_pts = pts[z <= low + 0.5]
        if len(base_pts) > 0:
            base = np.mean(base_pts, axis=0); valid[sel] = True
        else:
            base = np.array([0, 0, 0])
        position[sel] = base
    return position - xyz, valid


def crop_tile(rows, inner_t, outer_t, n_feat, inner_square_edge_length):
    """ONE tile of the loop of tile_generate_and_save (data_preparation.py:391-480) + TreeDataset.__getitem__ / collate_fn on it: `rows` =
    float32 [N, 4 + F] (x, y, z, label, features) of the whole plot, `inner_t` / `outer_t` = float64 (xmin, xmax, ymin, ymax).  Returns the
    batch dict (numpy) or None when the inner square holds no point (the reference drops such tiles, :412-427)."""
    x, y = rows[:, 0], rows[:, 1]
    o32 = np.asarray(outer_t).astype(np.float32)                          # torch compares against 0-dim float64 tensors in float32
    chunk = rows[(x >= o32[0]) & (x <= o32[1]) & (y >= o32[2]) & (y <= o32[3])]
    cx, cy = chunk[:, 0].astype(np.float64), chunk[:, 1].astype(np.float64)
    if not ((cx >= inner_t[0]) & (cx < inner_t[1]) & (cy > inner_t[2]) & (cy <= inner_t[3])).any():
        return None
    i32 = np.asarray(inner_t).astype(np.float32)
    c_x = np.round((i32[0] + i32[1]) / 2, 6); c_y = np.round((i32[2] + i32[3]) / 2, 6)      # float32 arithmetic
    center_row = np.concatenate([np.array([c_x, c_y, 0, 0]), np.zeros(n_feat)]).reshape(1, -1)
    chunk = (chunk.astype(np.float64) - center_row).astype(np.float32)
    xyz = chunk[:, :3]; inst = chunk[:, 3].astype(np.int32); feat = chunk[:, 4:]
    center = np.array([c_x, c_y, 0])                                  # what the npz holds (float64 of the float32 values)
    sem = np.where(inst == NON_TREE_RAW, NON_TREE_DS, TREE_DS).astype(np.float64)
    off, valid = offset_labels(xyz, inst, sem)
    m_inner = np.linalg.norm(xyz[:, :-1], ord=np.inf, axis=1) <= inner_square_edge_length / 2
    not_ignore = inst != IGNORE_RAW
    return dict(coords=xyz, input_feats=feat, batch_ids=np.zeros(len(xyz), np.int64), semantic_labels=sem.astype(np.int64),
                instance_labels=inst.astype(np.int64), masks_inner=m_inner,
                masks_off=m_inner & not_ignore & (sem != NON_TREE_DS) & valid, masks_sem=m_inner & not_ignore,
                offset_labels=off.astype(np.float32), batch_size=1,
                centers=(np.ones_like(xyz) * center).astype(np.float32))


def plot_tiles(points, labels, feats, inner_edge, outer_edge, stride, inner_square_edge_length):
    """points f32[N,3], labels f32[N], feats f32[N,F] -> list of batch dicts (numpy), one per kept tile, in the
    reference's tile order (the index i of `<plot>_<i>.npz`)."""
    points = np.asarray(points, np.float32); labels = np.asarray(labels, np.float32); feats = np.asarray(feats, np.float32)
    x_range = (points[:, 0].min(), points[:, 0].max()); y_range = (points[:, 1].min(), points[:, 1].max())
    inner, outer = tile_grid(x_range, y_range, inner_edge, outer_edge, stride)
    rows = np.hstack([points, labels[:, None], feats])                    # float32 [N, 4 + F]
    out = []
    for t in range(len(inner)):
        b = crop_tile(rows, inner[t], outer[t], feats.shape[1], inner_square_edge_length)
        if b is not None:
            out.append(b)
    return out
