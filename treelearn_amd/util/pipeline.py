"""Tile loop and instance grouping -- mirrors reference tree_learn/util/pipeline.py
(`get_pointwise_preds` :79-109, `get_instances` :145-169, `group_dbscan` :173-180,
`make_labels_consecutive` :195-206) with the device work on the HIP library."""
import numpy as np
import torch

from .. import ops


_GPU_KEYS = ("coords", "input_feats", "batch_ids", "masks_inner")


def _to_device_async(batch, stream):
    """Start the H2D copies of the tensors the forward needs on `stream` (non-blocking when the batch is pinned,
    as the reference DataLoader's pin_memory=True makes it; util/train.py:140)."""
    out = dict(batch)
    with torch.cuda.stream(stream):
        for k in _GPU_KEYS:
            v = batch.get(k)
            if torch.is_tensor(v) and not v.is_cuda:
                out[k] = v.cuda(non_blocking=True)
        ev = torch.cuda.Event(); ev.record(stream)
    return out, ev


def get_pointwise_preds(model, dataloader, config, logger=None, return_backbone_feats=True):
    """Per tile: forward, keep only `masks_inner` rows (filtered ON THE DEVICE, then one small D2H
    instead of the reference's three full-tile `.cpu()` copies), `coords += centers`, and skip tiles
    whose forward raises "... reach zero!!! ..." (pipeline.py:91-97).  The next tile's H2D copy runs on a
    side stream while the current tile computes."""
    outs = [[] for _ in range(8)]
    use_gpu = torch.cuda.is_available()
    copy_stream = torch.cuda.Stream() if use_gpu else None
    vs = getattr(config, 'voxel_size', None) if not isinstance(config, dict) else config.get('voxel_size')
    with torch.no_grad():
        model.eval()
        two_phase = use_gpu and hasattr(model, "prepare") and hasattr(model, "infer")

        def stage(b):
            """H2D on the copy stream, then (two-phase models) voxel hashing + rulebooks on the model's side stream;
            a tile whose U-Net would collapse raises here and is reported as skipped."""
            gb, ev = _to_device_async(b, copy_stream) if use_gpu else (b, None)
            if not two_phase:
                return gb, ev, None, None
            try:
                torch.cuda.current_stream().wait_event(ev)
                return gb, None, model.prepare(gb), None
            except Exception as e:                                     # noqa: BLE001
                return gb, None, None, e

        it = iter(dataloader)
        nxt = next(it, None)
        staged = stage(nxt) if nxt is not None else None
        while nxt is not None:
            batch, (gbatch, ev, handle, err) = nxt, staged
            gbatch['voxel_size'] = vs
            try:
                if err is not None:
                    raise err
                if two_phase:
                    output = model.infer(handle)                       # convs of this tile go on the main stream ...
                else:
                    if ev is not None:
                        torch.cuda.current_stream().wait_event(ev)
                    output = model(gbatch, return_loss=False)
            except Exception as e:                                     # noqa: BLE001
                if "reach zero!!!" in str(e):
                    if logger:
                        logger.info('Error in forward pass due to axis size collapse to zero during contraction of U-Net. '
                                    'If this does not happen too often, the results should not be influenced.')
                    nxt = next(it, None)
                    staged = stage(nxt) if nxt is not None else None
                    continue
                raise
            nxt = next(it, None)
            staged = stage(nxt) if nxt is not None else None           # ... while the next tile's geometry is built
            dev = output['offset_predictions'].device
            m_dev = gbatch['masks_inner'].to(dev)
            idx = torch.nonzero(m_dev).squeeze(1)                      # one small sync; 4-5 % of the rows survive
            off = output['offset_predictions'].index_select(0, idx).cpu()
            sem = output['semantic_prediction_logits'].index_select(0, idx).cpu()
            bb = output['backbone_feats']
            bb = bb.index_select(0, idx).cpu() if bb is not None else torch.zeros((idx.shape[0], 0))
            ci = idx.cpu()
            sel = lambda t: t.index_select(0, ci) if not t.is_cuda else t.index_select(0, idx).cpu()   # noqa: E731
            outs[0].append(sem); outs[1].append(sel(batch['semantic_labels']))
            outs[2].append(off); outs[3].append(sel(batch['offset_labels']))
            outs[4].append(sel(batch['coords']) + sel(batch['centers'])); outs[5].append(sel(batch['instance_labels']))
            outs[6].append(bb); outs[7].append(sel(batch['input_feats']))
    if not outs[0]:                  # every tile skipped (the reference would fail in torch.cat here)
        return tuple(np.zeros((0,), np.float32) for _ in outs)
    return tuple(torch.cat(o, 0).numpy() for o in outs)


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
