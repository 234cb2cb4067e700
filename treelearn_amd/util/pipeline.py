"""Tile loop and instance grouping -- mirrors reference tree_learn/util/pipeline.py
(`get_pointwise_preds` :79-109, `get_instances` :145-169, `group_dbscan` :173-180,
`make_labels_consecutive` :195-206) with the device work on the HIP library."""
import contextlib
import os

import numpy as np
import torch

from .. import ops


_GPU_KEYS = ("coords", "input_feats", "batch_ids", "masks_inner")


_STREAMS = []


def _compute_streams(n):
    """The tile loop's compute streams, created once per process: the caching allocator keeps one pool per stream, so fresh
    streams on every call would re-allocate every tile-sized buffer from the driver (measured: 80 instead of 8 ms per tile)."""
    while len(_STREAMS) < n:
        _STREAMS.append(torch.cuda.Stream())
    return _STREAMS[:n]


class _H2DRing:
    """Persistent device staging buffers for the tiles a loop has in flight.  `tensor.cuda()` per tile allocates on the copy stream and the
    blocks come back through record_stream events, which in practice meant a hipMalloc / hipFree pair -- both synchronise the device -- per
    tile and key: a loop over HOST-resident 40 m tiles took 36-51 ms per tile against 7.2 ms for device-resident ones (profiles/r5_final/
    pcie_inclusive.txt).  Here every slot owns its buffers (grown on demand, kept for the life of the process); a slot is re-used for tile
    k + depth only after the event that marks tile k's read-back (`release`)."""

    def __init__(self, depth, stream):
        self.depth, self.stream = depth, stream
        self.slots = [dict() for _ in range(depth)]
        self.free_ev = [None] * depth
        self.i = 0

    def stage(self, batch):
        """-> (batch dict with the forward's tensors on the device, event of the copies, slot index or None when nothing was copied)"""
        out = dict(batch)
        todo = [k for k in _GPU_KEYS if torch.is_tensor(batch.get(k)) and not batch[k].is_cuda]
        with torch.cuda.stream(self.stream):
            if not todo:
                ev = torch.cuda.Event(); ev.record(self.stream)
                return out, ev, None
            s = self.i % self.depth
            self.i += 1
            if self.free_ev[s] is not None:
                self.stream.wait_event(self.free_ev[s])
                self.free_ev[s] = None
            slot = self.slots[s]
            for k in todo:
                v = batch[k]
                n = v.shape[0]
                buf = slot.get(k)
                if buf is None or buf.shape[0] < n or buf.dtype != v.dtype or buf.shape[1:] != v.shape[1:]:
                    buf = slot[k] = torch.empty((int(n * 1.2) + 64,) + tuple(v.shape[1:]), dtype=v.dtype, device="cuda")
                view = buf[:n]
                view.copy_(v, non_blocking=True)                       # asynchronous when the batch is pinned (the reference DataLoader's pin_memory=True)
                view._tl_ring = True
                out[k] = view
            ev = torch.cuda.Event(); ev.record(self.stream)
        return out, ev, s

    def release(self, s, event):
        if s is not None:
            self.free_ev[s] = event


_RINGS = {}


def _h2d_ring(depth, stream):
    key = (torch.cuda.current_device(), depth)
    r = _RINGS.get(key)
    if r is None:
        r = _RINGS[key] = _H2DRing(depth, stream)
    r.stream = stream
    return r


def _cat_rows_numpy(parts):
    """torch.cat(parts, 0).numpy(), written into memory numpy allocated: numpy asks for transparent huge pages on large blocks (madvise), torch's
    CPU allocator does not, and with 4 KB pages the page faults of a fresh result cost more than the copy (measured on the GPU box's host: 12.5 MB
    of results per tile, 0.8 ms to fault in + 1.2 ms to unmap, against a 7.3 ms forward)."""
    if any(p.dtype != parts[0].dtype or p.shape[1:] != parts[0].shape[1:] for p in parts):
        return torch.cat(parts, 0).numpy()                             # mixed tiles: torch's promotion rules decide, as in the reference's torch.cat
    out = np.empty((sum(p.shape[0] for p in parts),) + tuple(parts[0].shape[1:]), dtype=parts[0][:0].numpy().dtype)
    if out.size:
        torch.cat(parts, 0, out=torch.from_numpy(out))
    return out


def get_pointwise_preds(model, dataloader, config, logger=None, return_backbone_feats=True, return_tile_rows=False,
                        keep_on_device=False):
    """Per tile: forward, keep only `masks_inner` rows (filtered ON THE DEVICE, then one packed D2H copy instead of the
    reference's full-tile `.cpu()` copies), `coords += centers`, and skip tiles whose forward raises
    "... reach zero!!! ..." (pipeline.py:91-97).

    Software-pipelined by one tile: the forward of tile i+1 is enqueued on the main stream before tile i is read back on a
    second stream (which waits only for the event recorded after forward i), so the synchronising read-back, the host-side
    bookkeeping and the launch overhead of the next tile hide behind GPU work.  The next tile's H2D copy runs on a third
    (copy) stream.  Building the next tile's geometry concurrently on a side stream was measured slower (the convs already
    fill the GPU and the geometry's host syncs stall the launch thread) and is not done.

    keep_on_device=True returns the 8 results as device tensors instead of numpy arrays (no D2H at all), for a consumer that
    continues on the GPU -- `postprocess.ensemble` accepts them."""
    outs = [[] for _ in range(8)]
    d2h_pending, deferred, d2h_bytes = [], [], [0]                     # asynchronous packed D2H copies (numpy results): events, deferred host arithmetic
    pinned_limit = int(os.environ.get("TL_D2H_PINNED_MB", "1024")) << 20  # pinned landing buffers held before they are folded into pageable memory
    tile_rows = []                                                     # (position in the iterable, inner rows) of every tile that produced output
    use_gpu = torch.cuda.is_available()
    copy_stream = torch.cuda.Stream() if use_gpu else None
    rb_stream = torch.cuda.Stream() if use_gpu else None
    main_stream = torch.cuda.current_stream() if use_gpu else None
    vs = getattr(config, 'voxel_size', None) if not isinstance(config, dict) else config.get('voxel_size')

    def read_back(pos, batch, gbatch, output, done, slot=None):
        n0 = sum(len(o) for o in outs[0])                              # (len() of a landing view is its row count: no read)
        _read_back_on(batch, gbatch, output, done)
        tile_rows.append((pos, sum(len(o) for o in outs[0]) - n0))
        if slot is not None:                                           # the tile's staging slot may take the next host tile once this read-back has run
            ev = torch.cuda.Event(); ev.record(rb_stream if done is not None else torch.cuda.current_stream())
            ring.release(slot, ev)

    def _read_back_on(batch, gbatch, output, done):
        if done is not None:
            with torch.cuda.stream(rb_stream):
                rb_stream.wait_event(done)
                return _read_back(batch, gbatch, output, True)
        return _read_back(batch, gbatch, output)

    def _read_back(batch, gbatch, output, done_on_other_stream=False):
        dev = output['offset_predictions'].device
        if done_on_other_stream and gbatch['masks_inner'].is_cuda and not getattr(gbatch['masks_inner'], "_tl_ring", False):
            gbatch['masks_inner'].record_stream(torch.cuda.current_stream())
        idx = torch.nonzero(gbatch['masks_inner'].to(dev)).squeeze(1)          # one small sync; 4-5 % of the rows survive
        n_in = idx.shape[0]
        src_of = lambda k: gbatch[k] if (torch.is_tensor(gbatch.get(k)) and gbatch[k].is_cuda) else batch[k]      # noqa: E731
        # the row list goes home FIRST (a small synchronous copy) when any per-point array has to be selected on the host: behind the packed
        # copy below it would wait for that copy
        ci = idx.cpu() if any(not src_of(k).is_cuda for k in ('semantic_labels', 'instance_labels', 'offset_labels', 'coords', 'centers', 'input_feats')) else None

        def rows(t):
            """Inner rows of a per-point array: gathered on the device if it lives there, else on the host."""
            if t.is_cuda:
                if done_on_other_stream and not getattr(t, "_tl_ring", False):
                    t.record_stream(torch.cuda.current_stream())       # allocated on a compute stream, read here on the read-back stream
                return t.index_select(0, idx)
            return t.index_select(0, ci)

        bb = output['backbone_feats'] if return_backbone_feats else None        # a model without the switch still returns them: not shipped
        # every device-side result goes home in ONE packed D2H copy (float columns) instead of one copy + sync per array
        cols = [rows(output['semantic_prediction_logits']).float(), rows(output['offset_predictions']).float()]
        if bb is not None:
            cols.append(rows(bb).float())
        on_dev = {}
        for k in ('offset_labels', 'coords', 'centers', 'input_feats'):
            src = src_of(k)
            if src.is_cuda:
                on_dev[k] = len(cols); cols.append(rows(src).float().reshape(n_in, -1))
        lab_dev = {}
        if not keep_on_device:
            for k in ('semantic_labels', 'instance_labels'):                    # device-resident integer labels ride along as bit patterns: exact for any dtype
                src = src_of(k)
                if src.is_cuda and src.dim() == 1 and src.element_size() in (4, 8):
                    lab_dev[k] = (len(cols), src.dtype); cols.append(rows(src).contiguous().view(torch.float32).reshape(n_in, -1))
        widths = [c.shape[1] for c in cols]
        packed = torch.cat(cols, 1)
        asynchronous = False
        if keep_on_device and packed.is_cuda:
            packed.record_stream(main_stream)                          # produced on the read-back stream, consumed on the main one
            host = packed
        elif packed.is_cuda:
            # asynchronous copy into pinned memory: the host goes on to enqueue the next tile.  Everything taken from `host` below is a VIEW of
            # the landing buffer; arithmetic on it is deferred to `settle()`, which first waits for the copies.
            host = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
            host.copy_(packed, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
            d2h_pending.append(ev)
            d2h_bytes[0] += host.numel() * 4
            asynchronous = True
        else:
            host = packed
        parts = list(torch.split(host, widths, dim=1))
        home = (lambda t: t.to(dev)) if keep_on_device else (lambda t: t.cpu() if t.is_cuda else t)   # noqa: E731
        get = lambda k: parts[on_dev[k]] if k in on_dev else home(rows(src_of(k)))                    # noqa: E731

        def put(i, make):
            """Result i of this tile is make(): now, or once the packed copy has landed when it reads the landing buffer."""
            if asynchronous:
                outs[i].append(None); deferred.append((i, len(outs[i]) - 1, make))
            else:
                outs[i].append(make())

        def lab(k):
            if k in lab_dev:
                c, dt = lab_dev[k]
                return parts[c].contiguous().view(dt).reshape(-1)
            return home(rows(src_of(k)))
        c_host = None if 'coords' in on_dev else get('coords')          # host-side selections happen now, while the copy is in flight
        z_host = None if 'centers' in on_dev else get('centers')
        outs[0].append(parts[0]); outs[2].append(parts[1])
        if 'semantic_labels' in lab_dev: put(1, lambda: lab('semantic_labels'))
        else: outs[1].append(lab('semantic_labels'))
        outs[3].append(get('offset_labels'))
        if c_host is not None and z_host is not None: outs[4].append(c_host + z_host)
        else: put(4, lambda: (parts[on_dev['coords']] if c_host is None else c_host) + (parts[on_dev['centers']] if z_host is None else z_host))
        if 'instance_labels' in lab_dev: put(5, lambda: lab('instance_labels'))
        else: outs[5].append(lab('instance_labels'))
        outs[6].append(parts[2] if bb is not None else torch.zeros((n_in, 0), device=parts[0].device)); outs[7].append(get('input_feats'))
        if d2h_bytes[0] > pinned_limit:
            settle(consolidate=True)

    def settle(consolidate=False):
        """Wait for the asynchronous D2H copies, run the arithmetic deferred on them; `consolidate` also moves what has arrived into pageable
        memory, which frees the pinned landing buffers of a long plot (thousands of tiles) for re-use."""
        for ev in d2h_pending:
            ev.synchronize()
        d2h_pending.clear()
        for i, j, make in deferred:
            outs[i][j] = make()
        deferred.clear()
        if consolidate:
            for i in range(8):
                outs[i] = [torch.from_numpy(_cat_rows_numpy(outs[i]))]
            d2h_bytes[0] = 0

    # Tiles in flight: consecutive tiles run on NF compute streams round-robin, so that the stretches of one forward that leave
    # most CUs idle (the 36 small launches of the deep levels, the geometry kernels and their two host syncs) are filled by the
    # big convs of another tile: measured 8.63 -> 7.90 (2) -> 7.69 (3) -> 7.96 (4) ms per 40 m tile (tools/dev_two_streams.py).
    # (four tiles in flight need eight hardware queues -- GPU_MAX_HW_QUEUES=8, set by the package at import when the HIP runtime has not started
    # yet; with the runtime's default of four the fourth tile's stream shares a queue and three in flight are faster: 7.69 vs 7.96 ms per tile)
    from .. import hw_queues
    nf = max(1, int(os.environ.get("TL_TILES_IN_FLIGHT", "4" if hw_queues() >= 8 else "3"))) if use_gpu else 1
    if os.environ.get("TL_LOOP_PIPELINE", "1") == "0":
        nf = 1
    cstreams = _compute_streams(nf) if (use_gpu and nf > 1) else []
    restore_bb = None
    if hasattr(model, "return_backbone_feats") and not return_backbone_feats:
        restore_bb = model.return_backbone_feats                        # skip the [N, 32] backbone output (never used downstream, reference
        model.return_backbone_feats = False                             # tools/pipeline/pipeline.py only saves it); result 6 is then [n, 0]
    try:
        with torch.no_grad():
            model.eval()
            if cstreams and hasattr(model, "ensure_plan"):
                model.ensure_plan()                                        # on the caller's stream, which every compute stream waits for
            it = iter(dataloader)
            nxt = next(it, None)
            ring = _h2d_ring(nf + 3, copy_stream) if use_gpu else None
            staged = ring.stage(nxt) if (nxt is not None and use_gpu) else (nxt, None, None)
            pending = []                                                   # tiles whose results are still on the device, oldest first
            pos = -1
            while nxt is not None:
                batch, (gbatch, ev, slot) = nxt, staged
                pos += 1
                gbatch['voxel_size'] = vs
                nxt = next(it, None)
                staged = ring.stage(nxt) if (nxt is not None and use_gpu) else (nxt, None, None)
                cs = cstreams[pos % nf] if cstreams else None
                if cs is not None:
                    cs.wait_stream(main_stream)                            # inputs produced on the caller's stream are visible
                try:
                    with (torch.cuda.stream(cs) if cs is not None else contextlib.nullcontext()):
                        if cs is not None:                                 # inputs allocated on the copy / tiler / caller's stream, consumed on this one
                            for v in gbatch.values():
                                if torch.is_tensor(v) and v.is_cuda and not getattr(v, "_tl_ring", False):
                                    v.record_stream(cs)
                        if ev is not None:
                            torch.cuda.current_stream().wait_event(ev)
                        if gbatch.get('_ready_event') is not None:          # device-resident tile produced on another stream (PlotTiler)
                            torch.cuda.current_stream().wait_event(gbatch['_ready_event'])
                        output = model(gbatch, return_loss=False)
                        done = None
                        if use_gpu:
                            done = torch.cuda.Event(); done.record()
                except Exception as e:                                     # noqa: BLE001
                    if "reach zero!!!" not in str(e):
                        raise
                    if logger:
                        logger.info('Error in forward pass due to axis size collapse to zero during contraction of U-Net. '
                                    'If this does not happen too often, the results should not be influenced.')
                    if slot is not None:
                        e2 = torch.cuda.Event(); e2.record(cs if cs is not None else torch.cuda.current_stream())
                        ring.release(slot, e2)
                    continue
                if os.environ.get("TL_LOOP_PIPELINE", "1") == "0":           # A/B switch: read every tile back right away
                    read_back(pos, batch, gbatch, output, None, slot)
                    continue
                pending.append((pos, batch, gbatch, output, done, slot))
                if len(pending) > nf:
                    read_back(*pending.pop(0))                             # the oldest tile comes home while nf younger ones compute
            for pnd in pending:
                read_back(*pnd)
            # the unit builder's assertion flag of a tile's forward is otherwise only seen by the NEXT forward on the same stream: after the last
            # tile, ask for it (every tile's results have been read back, so nothing waits here)
            ex = getattr(getattr(model, "_plan", None), "_exec", None)
            if ex:
                ex.check()
    finally:
        if restore_bb is not None:                                    # also when a tile raised: later callers must not inherit the flag
            model.return_backbone_feats = restore_bb
    if cstreams:
        for cs in cstreams:
            main_stream.wait_stream(cs)
    if not outs[0]:                  # every tile skipped (the reference would fail in torch.cat here)
        res = tuple(np.zeros((0,), np.float32) for _ in outs)
    elif keep_on_device:
        if use_gpu:
            main_stream.wait_stream(rb_stream)
        res = tuple(torch.cat(o, 0) for o in outs)
    else:
        settle()                                                       # the views appended above point into pinned buffers that are landing
        res = tuple(_cat_rows_numpy(o) for o in outs)
    return (res, tile_rows) if return_tile_rows else res


def get_instances_device(coords, offset, semantic_prediction_logits, grouping_cfg, verticality_feat, tree_class_in_dataset,
                         non_trees_label_in_grouping, not_assigned_label_in_grouping, start_num_preds):
    """`get_instances` for DEVICE tensors (the ensemble's outputs left in HBM): the masks, the shifted coordinates and the label
    assembly stay on the GPU; only the labels of the few hundred thousand grouped points cross PCIe for the size filter.  Same
    arithmetic as the numpy form (float32 softmax, the reference's threshold tests): identical ids.  Returns an int64 device tensor."""
    g = (lambda k: grouping_cfg[k]) if isinstance(grouping_cfg, dict) else (lambda k: getattr(grouping_cfg, k))
    cluster_coords = (coords + offset)[:, :3]
    probs = semantic_prediction_logits.float().softmax(dim=-1)
    tree_mask = probs[:, tree_class_in_dataset] >= g('tree_conf_thresh')
    mask = tree_mask & (verticality_feat > g('tau_vert')) & (offset[:, 2].abs() < g('tau_off'))
    ind = torch.nonzero(mask).squeeze(1)
    xy = cluster_coords.index_select(0, ind)[:, :2].contiguous()
    predictions = torch.full((coords.shape[0],), float(non_trees_label_in_grouping), dtype=torch.float64, device=coords.device)
    predictions[tree_mask] = not_assigned_label_in_grouping
    if g('use_hdbscan'):
        inst = group_hdbscan(xy, g('tau_min'), not_assigned_label_in_grouping, start_num_preds)
    else:
        inst = group_dbscan(xy, g('tau_group'), g('tau_min'), not_assigned_label_in_grouping, start_num_preds)
    predictions[ind] = torch.from_numpy(np.asarray(inst)).to(predictions.device, torch.float64)
    return predictions.to(torch.int64)


def make_labels_consecutive(labels, start_num):
    """Relabel to start_num.. in ascending order of the original labels; also the new->old map."""
    palette = np.unique(labels)
    new = np.searchsorted(palette, labels) + start_num
    return new, {i + start_num: orig for i, orig in enumerate(palette)}


def _filter_small(labels, npoint_thr, not_assigned, start_num):
    nums, counts = np.unique(labels, return_counts=True)
    valid = nums[(counts >= npoint_thr) & (nums != -1)]
    ok = np.isin(labels, valid)
    out = np.full(len(labels), not_assigned, dtype=np.int64)
    if ok.any():
        out[ok], _ = make_labels_consecutive(labels[ok], start_num)
    return out


def group_dbscan(cluster_coords, radius, npoint_thr, not_assigned_label_in_grouping, start_num_preds):
    """DBSCAN(eps=radius, min_samples=2) == connected components of the eps-graph; on the GPU."""
    from ..cluster import dbscan_min2
    labels = dbscan_min2(cluster_coords, radius)
    return _filter_small(labels, npoint_thr, not_assigned_label_in_grouping, start_num_preds)


def group_hdbscan(cluster_coords, npoint_thr, not_assigned_label_in_grouping, start_num_preds):
    from ..cluster import hdbscan
    labels = hdbscan(cluster_coords, npoint_thr)
    return _filter_small(labels, npoint_thr, not_assigned_label_in_grouping, start_num_preds)


def get_instances(coords, offset, semantic_prediction_logits, grouping_cfg, verticality_feat, tree_class_in_dataset,
                  non_trees_label_in_grouping, not_assigned_label_in_grouping, start_num_preds):
    g = (lambda k: grouping_cfg[k]) if isinstance(grouping_cfg, dict) else (lambda k: getattr(grouping_cfg, k))
    cluster_coords = (coords + offset)[:, :3]
    probs = torch.from_numpy(semantic_prediction_logits).float().softmax(dim=-1)
    tree_mask = (probs[:, tree_class_in_dataset] >= g('tree_conf_thresh')).numpy()
    mask = tree_mask & (verticality_feat > g('tau_vert')) & (np.abs(offset[:, 2]) < g('tau_off'))
    ind = np.where(mask)[0]
    xy = cluster_coords[ind][:, :2]
    predictions = non_trees_label_in_grouping * np.ones(len(cluster_coords))
    predictions[tree_mask] = not_assigned_label_in_grouping
    if g('use_hdbscan'):
        inst = group_hdbscan(xy, g('tau_min'), not_assigned_label_in_grouping, start_num_preds)
    else:
        inst = group_dbscan(xy, g('tau_group'), g('tau_min'), not_assigned_label_in_grouping, start_num_preds)
    predictions[ind] = inst
    return predictions.astype(np.int64)
