"""Device-resident inference tiling -- SURVEY.md 8f #3.

Mirrors, for the pipeline's configuration (plot_corners=None, no tile denoising), the chain
  SampleGenerator.tile_generate_and_save  (reference tree_learn/util/data_preparation.py:333-494)
  -> <plot>_<i>.npz -> TreeDataset.__getitem__ (test mode) -> collate_fn, batch size 1
     (tree_learn/dataset/dataset.py:34-76,167-226)
without the npz round trip: the voxelised plot stays in HBM, every tile is one `tl_tile_crop` call (box test,
stable compaction, centring, labels and masks fused), and the batch dict handed to the tile loop
(`util/pipeline.get_pointwise_preds`) holds device tensors.  Tile order, tile skipping (no point inside the inner
square) and every dtype decision are the reference's (golden G11).

Offset labels (dataset.py:111-140) depend on numpy's `np.partition(z, 10)[3]` -- an implementation-defined element
among the ten lowest points -- and on float32 summation order, so for bit-identical labels they are derived on the
host from the cropped tile exactly as the reference's DataLoader worker does (`offset_labels="host"`, the default).
Pure inference does not use them (`get_instances` reads coordinates, logits, offsets and the verticality
feature only): `offset_labels="none"` skips the D2H copy and returns zeros / an all-false `masks_off`.
"""
import ctypes

import numpy as np
import torch

from .. import _hip

IGNORE_LABEL, NON_TREE_LABEL = -1, 0            # raw-data conventions, dataset.py:7-8
NON_TREE_CLASS, TREE_CLASS = 1, 0               # dataset.py:9-10


def tile_grid(x_range, y_range, inner_edge, outer_edge, stride):
    """Inner and outer square extensions, float64 [T,4] = (x0, x1, y0, y1), tiles row-major from the top-left corner
    (data_preparation.py:362-389: the plot is padded by 1.5 outer edges, the inner edge is adapted so that an integer
    number of squares fits, then the squares are laid out every `stride` inner edges).  The plot ranges are float32
    (the voxelised plot is stored as float32) and numpy keeps the whole lay-out arithmetic in float32 -- Python
    floats are weak scalars -- before the result is widened and rounded to 5 decimals; restated the same way here."""
    f32 = np.float32
    lo = [np.round(f32(r[0]) - f32(1.5 * outer_edge), 2) for r in (x_range, y_range)]
    hi = [np.round(f32(r[1]) + f32(1.5 * outer_edge), 2) for r in (x_range, y_range)]
    span = [h - l - f32(2 * outer_edge) for l, h in zip(lo, hi)]
    n_fit = [int(np.round(sp / f32(inner_edge))) for sp in span]
    edge = [np.round(sp / f32(k), 5) for sp, k in zip(span, n_fit)]
    ncols, nrows = (int((k - 1) / stride + 1) for k in n_fit)
    sj = (stride * np.arange(ncols)).astype(f32); sj1 = (stride * np.arange(ncols) + 1).astype(f32)
    si = (stride * np.arange(nrows)).astype(f32); si1 = (stride * np.arange(nrows) + 1).astype(f32)
    left = lo[0] + f32(outer_edge); top = hi[1] - f32(outer_edge)
    x0 = left + sj * edge[0]; x1 = left + sj1 * edge[0]
    y0 = top - si1 * edge[1]; y1 = top - si * edge[1]
    inner = np.stack([np.tile(x0, nrows), np.tile(x1, nrows), np.repeat(y0, ncols), np.repeat(y1, ncols)], axis=1).astype(np.float64)
    inner = np.round(inner, 5)
    outer = inner + np.array([-outer_edge, outer_edge, -outer_edge, outer_edge])
    return inner, outer


def _offset_labels_host(xyz, inst, sem):
    """dataset.py:111-140 on the cropped tile (numpy, as in the reference's DataLoader worker)."""
    position = np.ones_like(xyz, dtype=np.float32)
    valid = np.zeros(len(xyz), dtype=bool)
    order = np.argsort(inst, kind="stable")
    bounds = np.flatnonzero(np.diff(inst[order])) + 1
    for sel in np.split(order, bounds):
        if sem[sel[0]] == NON_TREE_CLASS:
            continue
        pts = xyz[sel]; z = pts[:, 2]
        low = np.partition(z, 10)[3] if len(z) > 11 else z.min()
        base_pts = pts[z <= low + 0.5]
        if len(base_pts):
            position[sel] = np.mean(base_pts, axis=0); valid[sel] = True
        else:
            position[sel] = 0.0
    return position - xyz, valid


class PlotTiler:
    """Holds the voxelised plot (points f32[N,3], labels f32[N], features f32[N,F]) on the device and yields the
    reference's inference tiles as batch dicts."""

    def __init__(self, points, labels, feats, device="cuda"):
        as_t = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))).to(device=device, dtype=torch.float32).contiguous()  # noqa: E731
        self.xyz = as_t(points); self.labels = as_t(labels).reshape(-1); self.feats = as_t(feats).reshape(len(self.xyz), -1)
        assert self.xyz.is_cuda, "PlotTiler needs the HIP library and a GPU"
        n = len(self.xyz)
        mn = self.xyz[:, :2].amin(0).cpu().numpy(); mx = self.xyz[:, :2].amax(0).cpu().numpy()      # get_ranges, data_preparation.py:497-508
        self.x_range, self.y_range = (mn[0], mx[0]), (mn[1], mx[1])
        F = self.feats.shape[1]; dev = self.xyz.device
        L = _hip.lib()
        self._ws = torch.empty(int(L.tl_tile_crop_ws_words(n)), dtype=torch.int32, device=dev)
        self._count = torch.empty(2, dtype=torch.int32, device=dev)
        self._stream = torch.cuda.Stream(device=dev)        # crops run beside the consumer's forward passes, not behind them
        self._buf = dict(coords=torch.empty((n, 3), dtype=torch.float32, device=dev), feats=torch.empty((n, max(F, 1)), dtype=torch.float32, device=dev),
                         inst=torch.empty(n, dtype=torch.int64, device=dev), sem=torch.empty(n, dtype=torch.int64, device=dev),
                         m_inner=torch.empty(n, dtype=torch.uint8, device=dev), m_sem=torch.empty(n, dtype=torch.uint8, device=dev))

    def crop(self, inner, outer, inner_square_edge_length):
        """One tile: (kept rows, rows in the inner square, centre f64[3]); the rows are in self._buf[...][:kept]."""
        box = _hip.TileBox()
        i32 = inner.astype(np.float32)
        cx = np.round((i32[0] + i32[1]) / 2, 6); cy = np.round((i32[2] + i32[3]) / 2, 6)          # float32 arithmetic, data_preparation.py:431-434
        box.outer[:] = [float(v) for v in outer.astype(np.float32)]
        box.inner[:] = [float(v) for v in inner]
        box.center[:] = [float(cx), float(cy)]
        box.half_inner = float(inner_square_edge_length) / 2
        b = self._buf; F = self.feats.shape[1]
        _hip.check(_hip.lib().tl_tile_crop(_hip.ptr(self.xyz), _hip.ptr(self.labels), _hip.ptr(self.feats), len(self.xyz), F, ctypes.byref(box),
                                           _hip.ptr(b["coords"]), _hip.ptr(b["feats"]), _hip.ptr(b["inst"]), _hip.ptr(b["sem"]),
                                           _hip.ptr(b["m_inner"]), _hip.ptr(b["m_sem"]), _hip.ptr(self._count), _hip.ptr(self._ws), _hip.stream()),
                   "tl_tile_crop")
        kept, n_inner = (int(v) for v in self._count.cpu())
        return kept, n_inner, np.array([float(cx), float(cy), 0.0])

    def tiles(self, inner_edge, outer_edge, stride, inner_square_edge_length, offset_labels="host"):
        """Generator of batch dicts (collate_fn's keys, batch size 1), in the reference's tile numbering."""
        assert offset_labels in ("host", "none")
        inner, outer = tile_grid(self.x_range, self.y_range, inner_edge, outer_edge, stride)
        F = self.feats.shape[1]
        main = torch.cuda.current_stream()
        self._stream.wait_stream(main)                         # the plot arrays may have just been produced on the caller's stream
        for t in range(len(inner)):                            # (one wait for the whole generator: a per-tile wait would put every crop behind the consumer's previous forward)
            batch = self.tile_batch(inner[t], outer[t], inner_square_edge_length, offset_labels, tile_index=t, sync_with_caller=False)
            if batch is not None:
                yield batch

    def tile_batch(self, inner, outer, inner_square_edge_length, offset_labels="host", tile_index=0, sync_with_caller=True):
        """ONE tile as a batch dict (collate_fn's keys, batch size 1) from its inner / outer square (x0, x1, y0, y1), or None when the inner
        square holds no point (data_preparation.py:412-427).  Random access for callers that own only some tiles of a plot (a rank of the
        sharded tile loop) or lay the squares out themselves.  `sync_with_caller=False`: the caller has already made the tiler's stream wait
        for whatever produced the plot arrays (`tiles()` does so once) -- the crop then does not queue behind the caller's stream."""
        assert offset_labels in ("host", "none")
        F = self.feats.shape[1]
        main = torch.cuda.current_stream()
        if sync_with_caller:
            self._stream.wait_stream(main)                     # the plot arrays may have just been produced on the caller's stream
        t = tile_index
        # The crop (and its one host sync for the row count) goes on the tiler's own stream: a consumer that pulls the next
        # tile before launching the current forward (util/pipeline.get_pointwise_preds) then never waits for its own convs.
        with torch.cuda.stream(self._stream):
            kept, n_inner, center = self.crop(np.asarray(inner, np.float64), np.asarray(outer, np.float64), inner_square_edge_length)
            if n_inner == 0:                               # data_preparation.py:412-427: tiles whose inner square is empty are dropped
                return None
            b = self._buf
            coords = b["coords"][:kept].clone(); inst = b["inst"][:kept].clone(); sem = b["sem"][:kept].clone()
            m_inner = b["m_inner"][:kept].bool(); m_sem = b["m_sem"][:kept].bool()
            if offset_labels == "host":
                off, valid = _offset_labels_host(coords.cpu().numpy(), inst.cpu().numpy(), sem.cpu().numpy())
                off_t = torch.from_numpy(off.astype(np.float32)).to(coords.device)
                m_off = m_sem & (sem != NON_TREE_CLASS) & torch.from_numpy(valid).to(coords.device)
            else:
                off_t = torch.zeros_like(coords); m_off = torch.zeros_like(m_sem)
            c32 = torch.from_numpy(center.astype(np.float32)).to(coords.device)
            batch = dict(coords=coords, input_feats=b["feats"][:kept, :F].clone(), batch_ids=torch.zeros(kept, dtype=torch.int64, device=coords.device),
                         semantic_labels=sem, instance_labels=inst, masks_inner=m_inner, masks_off=m_off, masks_sem=m_sem,
                         offset_labels=off_t, batch_size=1, centers=c32.expand(kept, 3).contiguous(), tile_index=t)
            ready = torch.cuda.Event(); ready.record(self._stream)
        for v in batch.values():
            if torch.is_tensor(v):
                v.record_stream(main)                      # allocated on the tiler's stream, consumed on the caller's
        batch["_ready_event"] = ready                      # consumers on another stream wait for this before reading the tile
        return batch
