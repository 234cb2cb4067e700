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


class _ResultSink:
    """Where the numpy results of the tile loop are assembled, by a worker thread, while the launch thread goes on with the next tiles.

    Per tile the launch thread hands over an event and a closure; the worker waits for the event (the tile's packed D2H copy into a pinned
    landing buffer), runs the closure (-> the tile's 8 host arrays: views of the landing buffer, host-side row selections, `coords + centers`)
    and copies them to the end of 8 growing numpy arrays.  What this buys (16 tiles of 40 m, 12.5 MB of results each): the copy into pageable
    memory and the page faults of that memory -- 1.2 ms per tile when done at the end of the loop, more than the D2H itself -- run beside the
    forwards instead of after them, and a landing buffer goes back to torch's pinned cache as soon as its tile is filed, so a plot of thousands
    of tiles holds a handful of them.  The arrays are numpy-allocated (numpy asks for transparent huge pages on big blocks, torch's CPU
    allocator does not: 4 KB faults doubled the cost) and over-allocated from the first tile's size x the number of tiles when the iterable
    has a length (x 1.5 growth otherwise); the results are views of them, trimmed by a copy only if more than a quarter would be wasted."""

    def __init__(self, n_tiles_hint=None):
        import queue
        import threading
        self.hint = n_tiles_hint
        self.bufs, self.n, self.err = None, 0, None
        self.q = queue.Queue()
        self.t = threading.Thread(target=self._run, name="tl-result-sink", daemon=True)
        self.t.start()

    def push(self, event, make):
        self.q.put((event, make))

    def _run(self):
        while True:
            job = self.q.get()
            if job is None:
                return
            if self.err is not None:
                continue                                                # drain: the launch thread re-raises in finish()
            try:
                event, make = job
                if event is not None:
                    event.synchronize()
                self._append([a if torch.is_tensor(a) else torch.from_numpy(np.asarray(a)) for a in make()])
            except BaseException as e:                                 # noqa: BLE001
                self.err = e

    def _append(self, arrays):
        n = arrays[0].shape[0]
        if self.bufs is None:
            cap = int(n * (self.hint if self.hint else 4) * 1.15) + 1024
            self.bufs = [np.empty((cap,) + tuple(a.shape[1:]), dtype=a[:0].numpy().dtype) for a in arrays]
        for i, a in enumerate(arrays):
            buf = self.bufs[i]
            dt = np.result_type(buf.dtype, a[:0].numpy().dtype)       # tiles of mixed types: numpy's promotion, as torch.cat's in the reference
            if self.n + n > buf.shape[0] or dt != buf.dtype or tuple(a.shape[1:]) != buf.shape[1:]:
                if tuple(a.shape[1:]) != buf.shape[1:]:
                    raise RuntimeError(f"tile results of different widths: {tuple(a.shape)} after {buf.shape}")
                grown = np.empty((max(int(buf.shape[0] * 1.5), self.n + n) + 1024,) + buf.shape[1:], dtype=dt)
                grown[:self.n] = buf[:self.n]
                buf = self.bufs[i] = grown
            if n:
                torch.from_numpy(buf[self.n:self.n + n]).copy_(a)
        self.n += n

    def finish(self):
        self.q.put(None)
        self.t.join()
        if self.err is not None:
            raise self.err
        if self.bufs is None:
            return None
        return tuple((b[:self.n] if self.n * 4 >= b.shape[0] * 3 else b[:self.n].copy()) for b in self.bufs)

    def abandon(self):
        if self.t.is_alive():
            self.q.put(None)
            self.t.join()


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
    sink = None                                                        # numpy results of device tensors: assembled by a worker thread (_ResultSink)
    tile_rows = []                                                     # (position in the iterable, inner rows) of every tile that produced output
    use_gpu = torch.cuda.is_available()
    if use_gpu and not keep_on_device and os.environ.get("TL_RESULT_SINK", "1") != "0":
        sink = _ResultSink(len(dataloader) if hasattr(dataloader, "__len__") else None)
    copy_stream = torch.cuda.Stream() if use_gpu else None
    rb_stream = torch.cuda.Stream() if use_gpu else None
    main_stream = torch.cuda.current_stream() if use_gpu else None
    vs = getattr(config, 'voxel_size', None) if not isinstance(config, dict) else config.get('voxel_size')

    def read_back(pos, batch, gbatch, output, done, slot=None):
        tile_rows.append((pos, _read_back_on(batch, gbatch, output, done)))
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
        src_of = lambda k: gbatch[k] if (torch.is_tensor(gbatch.get(k)) and gbatch[k].is_cuda and gbatch[k].device == dev) else batch[k]      # noqa: E731
        to_sink = sink is not None                                     # (a tile whose results already live on the host goes through the sink as well: one order)
        ci = [None]                                                    # the row list on the host: fetched at most once, by whoever selects host rows

        def rows(t):
            """Inner rows of a per-point array: gathered on the device if it lives there, else on the host."""
            if t.is_cuda:
                if done_on_other_stream and not getattr(t, "_tl_ring", False):
                    t.record_stream(torch.cuda.current_stream())       # allocated on a compute stream, read here on the read-back stream
                return t.index_select(0, idx)
            if ci[0] is None:
                ci[0] = idx.cpu()
            return t.index_select(0, ci[0])

        bb = output['backbone_feats'] if return_backbone_feats else None        # a model without the switch still returns them: not shipped
        # every device-side result goes home in ONE packed D2H copy (float columns) instead of one copy + sync per array
        cols = [rows(output['semantic_prediction_logits']).float(), rows(output['offset_predictions']).float()]
        if bb is not None:
            cols.append(rows(bb).float())
        width = lambda t: int(np.prod(t.shape[1:], dtype=np.int64))     # noqa: E731  (explicit widths: a tile may have no inner row, and reshape(0, -1) is ambiguous)
        on_dev = {}
        for k in ('offset_labels', 'coords', 'centers', 'input_feats'):
            src = src_of(k)
            if src.is_cuda:
                on_dev[k] = len(cols); cols.append(rows(src).float().reshape(n_in, width(src)))
        bits = {}                                                      # columns riding along as bit patterns: (first column, carrier type, final type, row shape)
        if to_sink:
            # the worker thread must not touch device memory (its current stream is not this one): whatever lives on the device rides in the block
            for k in ('semantic_labels', 'instance_labels'):
                src = src_of(k)
                if src.is_cuda:
                    carrier = src.dtype if src.element_size() in (4, 8) else torch.float64           # (every 1- / 2-byte type is exact in a double)
                    bits[k] = (len(cols), carrier, src.dtype, tuple(src.shape[1:]))
                    cols.append(rows(src).to(carrier).contiguous().view(torch.float32).reshape(n_in, width(src) * (torch.empty((), dtype=carrier).element_size() // 4)))
            if any(not src_of(k).is_cuda for k in ('semantic_labels', 'instance_labels', 'offset_labels', 'coords', 'centers', 'input_feats')):
                bits['_idx'] = (len(cols), torch.int64, torch.int64, ())                              # the worker selects the host rows
                cols.append(idx.view(torch.float32).reshape(n_in, 2))
        widths = [c.shape[1] for c in cols]
        packed = torch.cat(cols, 1)
        if keep_on_device and packed.is_cuda:
            packed.record_stream(main_stream)                          # produced on the read-back stream, consumed on the main one
        ev = None
        if to_sink and packed.is_cuda:
            # asynchronous copy into a pinned landing buffer; everything that READS it runs on the sink's worker thread, after the event
            host = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
            host.copy_(packed, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
        else:
            host = packed if (keep_on_device or not packed.is_cuda) else packed.cpu()

        def finish():
            """The tile's 8 results from the landed block (runs on the worker thread when the results go to the sink).
            (clone, not contiguous(): a one-row column slice IS contiguous and keeps its odd storage offset, which an 8-byte view refuses)"""
            parts = list(torch.split(host, widths, dim=1))
            unbits = lambda k: parts[bits[k][0]].clone(memory_format=torch.contiguous_format).view(bits[k][1]).reshape((n_in,) + bits[k][3]).to(bits[k][2])   # noqa: E731
            if '_idx' in bits:
                ci[0] = unbits('_idx')
            home = (lambda t: t.to(dev)) if keep_on_device else (lambda t: t.cpu() if t.is_cuda else t)   # noqa: E731
            get = lambda k: parts[on_dev[k]] if k in on_dev else home(rows(src_of(k)))                    # noqa: E731
            lab = lambda k: unbits(k) if k in bits else home(rows(src_of(k)))                             # noqa: E731
            return [parts[0], lab('semantic_labels'), parts[1], get('offset_labels'), get('coords') + get('centers'), lab('instance_labels'),
                    parts[2] if bb is not None else torch.zeros((n_in, 0), device=parts[0].device), get('input_feats')]

        if to_sink:
            sink.push(ev, finish)
        else:
            for o, r in zip(outs, finish()):
                o.append(r)
        return n_in

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
    except BaseException:
        if sink is not None:
            sink.abandon()
        raise
    finally:
        if restore_bb is not None:                                    # also when a tile raised: later callers must not inherit the flag
            model.return_backbone_feats = restore_bb
    if cstreams:
        for cs in cstreams:
            main_stream.wait_stream(cs)
    res = sink.finish() if sink is not None else None                  # (waits for the last tiles' copies; raises what the worker met)
    if (sink is not None and res is None) or (sink is None and not outs[0]):  # every tile skipped (the reference would fail in torch.cat here)
        res = tuple(np.zeros((0,), np.float32) for _ in outs)
    elif keep_on_device:
        if use_gpu:
            main_stream.wait_stream(rb_stream)
        res = tuple(torch.cat(o, 0) for o in outs)
    elif sink is None:
        res = tuple(torch.cat(o, 0).numpy() for o in outs)
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
