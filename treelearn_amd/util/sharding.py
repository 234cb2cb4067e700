"""Tile sharding over the GPUs of one node (the reference has no distributed code).

Tiles are independent forward passes (reference tree_learn/util/pipeline.py:83-88), so the tile loop
shards with NO data-path collective: one process per GPU, static longest-processing-time assignment
by point count, each rank materialises and runs ONLY its own tiles.  The exchange afterwards is

    1. one all-gather of the per-rank row counts                         (world x i64)
    2. one padded all-gather of the packed inner-square records          ([rows, W] f32, 4-5 % of the points)
    3. (segment_plot_sharded) one broadcast of the i32 instance ids      -- the north star's "instance-id gather"

with the records kept on the device from the tile loop to the collective (no host round trip); the order of
operations after the gather -- ensemble, grouping, k-NN fill -- is the reference's
(tools/pipeline/pipeline.py:70-94).
"""
import numpy as np
import torch
import torch.distributed as dist

# columns of the packed record, in the order of get_pointwise_preds' 8 results; integer columns travel bit-cast (i32 in an f32 lane)
_INT_RESULTS = (1, 5)                       # semantic_labels, instance_labels


def assign_tiles(n_points, world_size):
    """LPT: biggest tile first onto the least-loaded rank (ties -> lowest rank).  Returns a list of
    index lists, one per rank, each in ascending tile order."""
    order = sorted(range(len(n_points)), key=lambda i: (-int(n_points[i]), i))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda q: (load[q], q))
        out[r].append(i); load[r] += int(n_points[i])
    return [sorted(x) for x in out]


def _bits_f32(t):
    """int64 labels -> f32 lanes holding the i32 bit pattern (lossless; labels are far below 2^31)."""
    return t.to(torch.int32).contiguous().view(torch.float32)


def _bits_i64(t):
    return t.contiguous().view(torch.int32).to(torch.int64)


def pack_records(res, tile_index):
    """The 8 per-point results of `get_pointwise_preds` (tensors, device or host) + the tile index of every row ->
    one [n, W] f32 matrix and the column widths needed to split it again."""
    n = res[0].shape[0]
    cols = [_bits_f32(tile_index).reshape(n, 1)]
    widths, is1d = [], []
    for a, t in enumerate(res):
        is1d.append(t.dim() == 1)
        t2 = t.reshape(n, -1)
        cols.append(_bits_f32(t2) if a in _INT_RESULTS else t2.float())
        widths.append(t2.shape[1])
    return torch.cat(cols, 1).contiguous(), widths, is1d


def unpack_records(packed, widths, is1d):
    parts = torch.split(packed[:, 1:], widths, dim=1)
    out = []
    for a, (p, one) in enumerate(zip(parts, is1d)):
        p = _bits_i64(p) if a in _INT_RESULTS else p
        out.append(p[:, 0] if one else p)
    return tuple(out)


class TileList:
    """Lazy tile source: `n_points[i]` is known up front (e.g. from PlotTiler's occupancy count or the npz headers),
    `make(i)` materialises tile i as a batch dict.  A rank only ever calls `make` for its own tiles."""

    def __init__(self, n_points, make):
        self.n_points = list(n_points); self.make = make

    def __len__(self):
        return len(self.n_points)


def _as_source(tiles):
    if isinstance(tiles, TileList):
        return tiles
    return TileList([t["coords"].shape[0] for t in tiles], lambda i: tiles[i])


def get_pointwise_preds_sharded(model, tiles, config, logger=None, group=None, device=None, return_device=False, return_backbone_feats=False):
    """`get_pointwise_preds` over `tiles` (a sequence of batch dicts or a TileList) sharded across the process group;
    every rank returns the full 8-array result in the original tile order.  Two collectives per plot.
    The record of a point is 1 + 2 + 1 + 3 + 3 + 3 + 1 + F f32 lanes = 60 bytes at F = 1; return_backbone_feats=True adds the 32
    backbone columns (188 bytes), which nothing downstream of the reference pipeline reads (tools/pipeline/pipeline.py only saves them)."""
    from .pipeline import get_pointwise_preds
    src = _as_source(tiles)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = assign_tiles(src.n_points, world)[rank]
    on_gpu = dist.get_backend(group) == "nccl"
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu"))
    # one pipelined pass over this rank's tiles, results left where the model put them (HBM on a GPU rank)
    res, rows = get_pointwise_preds(model, (src.make(i) for i in mine), config, logger, return_backbone_feats=return_backbone_feats,
                                    return_tile_rows=True, keep_on_device=True)
    res = tuple(torch.as_tensor(r).to(dev) for r in res)
    n_local = res[0].shape[0]
    tidx = torch.empty(n_local, dtype=torch.int64)
    off = 0
    for pos, n in rows:
        tidx[off:off + n] = mine[pos]; off += n
    packed, widths, is1d = (None, None, None) if n_local == 0 else pack_records(res, tidx.to(dev))
    # ranks with no rows learn the record layout from the others through the payload width: widths are a function of the
    # model (2, 1, 3, 3, 3, 1, C, F); exchange them inside collective 1's message instead of a third collective
    meta = torch.zeros(1 + 2 * 8, dtype=torch.int64, device=dev)
    if n_local:
        meta[0] = n_local
        meta[1:9] = torch.tensor(widths, dtype=torch.int64); meta[9:17] = torch.tensor([int(b) for b in is1d], dtype=torch.int64)
    metas = torch.empty(world * meta.numel(), dtype=torch.int64, device=dev)       # flat: gloo only takes the concatenated form
    dist.all_gather_into_tensor(metas, meta, group=group)                          # collective 1 (counts + layout)
    metas = metas.view(world, -1).tolist()
    counts = [m[0] for m in metas]
    if max(counts) == 0:
        empty = tuple(np.zeros((0,), np.float32) for _ in range(8))
        return tuple(torch.from_numpy(e) for e in empty) if return_device else empty
    ref = next(m for m in metas if m[0] > 0)
    widths, is1d = ref[1:9], [bool(b) for b in ref[9:17]]
    W = 1 + sum(widths)
    get_pointwise_preds_sharded.last_record_width = W             # f32 lanes per gathered point (tests assert the 60-byte record)
    m = max(counts)
    pad = torch.zeros((m, W), dtype=torch.float32, device=dev)
    if n_local:
        pad[:n_local] = packed
    buf = torch.empty(world * m * W, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(buf, pad.view(-1), group=group)                    # collective 2 (payload)
    buf = buf.view(world * m, W)
    allrec = buf if all(c == m for c in counts) else torch.cat([buf[r * m:r * m + c] for r, c in enumerate(counts)], 0)
    # single-GPU order = ascending tile index, rows of a tile in tile order: one stable sort of the tile column
    order = torch.sort(allrec[:, 0].contiguous().view(torch.int32), stable=True).indices
    allrec = allrec.index_select(0, order)
    out = unpack_records(allrec, widths, is1d)
    return out if return_device else tuple(o.cpu().numpy() for o in out)


def segment_plot_sharded(model, tiles, config, grouping_cfg, group=None, device=None, logger=None,
                         ensemble_fn=None, instances_fn=None, fill_fn=None,
                         tree_class=0, non_trees_label=0, not_assigned_label=-1, start_num_preds=1):
    """Whole-plot inference as the reference orders it (tools/pipeline/pipeline.py:70-94) with the tile loop sharded:
    tile loop (per rank) -> record gather (2 collectives) -> ensemble -> get_instances -> k-NN fill on rank 0 ->
    broadcast of the i32 instance ids (collective 3).  Returns (coords, instance_preds) as numpy on every rank.

    `ensemble_fn` / `instances_fn` / `fill_fn` default to the HIP implementations (util.postprocess / util.pipeline);
    CPU tests inject restatements."""
    from . import postprocess as pp
    from .pipeline import get_instances, get_instances_device
    rank = dist.get_rank(group)
    src_rank = dist.get_global_rank(group, 0) if group is not None else 0          # broadcast takes GLOBAL ranks; group rank 0 does the grouping
    on_gpu = dist.get_backend(group) == "nccl"
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu"))
    device_chain = on_gpu and ensemble_fn is None and instances_fn is None and fill_fn is None     # the HIP implementations: stay in HBM
    ensemble_fn = ensemble_fn or pp.ensemble
    instances_fn = instances_fn or get_instances
    fill_fn = fill_fn or pp.assign_remaining_points_nearest_neighbor
    sem, seml, off, offl, coords, instl, bb, infeat = get_pointwise_preds_sharded(model, tiles, config, logger, group, dev, return_device=on_gpu)
    # ensemble is replicated (deterministic on identical inputs: every rank knows the ensembled point count without a collective)
    if device_chain:
        # gather -> ensemble -> grouping -> k-NN fill -> broadcast without leaving the device: only the grouped points' labels (size
        # filter) and, at the very end, the results cross PCIe
        e_coords, e_sem, _, e_off, _, _, _, e_infeat = ensemble_fn(coords, sem, seml, off, offl, instl, bb, infeat, return_device=True)
        n = e_coords.shape[0]
        ids = torch.empty(n, dtype=torch.int32, device=dev)
        if rank == 0:
            inst = get_instances_device(e_coords, e_off, e_sem, grouping_cfg, e_infeat.reshape(n, -1)[:, -1], tree_class, non_trees_label,
                                        not_assigned_label, start_num_preds)
            tree = inst != non_trees_label
            it = inst[tree]
            if it.numel() and bool((it != not_assigned_label).any()) and bool((it == not_assigned_label).any()):
                inst[tree] = pp.assign_remaining_points_nearest_neighbor_device((e_coords + e_off)[tree], it, not_assigned_label)
            ids.copy_(inst.to(torch.int32))
        dist.broadcast(ids, src=src_rank, group=group)                                   # collective 3: the instance ids
        return e_coords.cpu().numpy(), ids.cpu().numpy().astype(np.int64)
    ens = ensemble_fn(coords, sem, seml, off, offl, instl, bb, infeat)
    e_coords, e_sem, _, e_off, _, _, _, e_infeat = (np.asarray(x.cpu()) if torch.is_tensor(x) else x for x in ens)
    n = len(e_coords)
    ids = torch.empty(n, dtype=torch.int32, device=dev)
    if rank == 0:
        inst = instances_fn(e_coords, e_off, e_sem, grouping_cfg, e_infeat.reshape(n, -1)[:, -1], tree_class, non_trees_label,
                            not_assigned_label, start_num_preds)
        tree = inst != non_trees_label
        if tree.any() and (inst[tree] != not_assigned_label).any() and (inst[tree] == not_assigned_label).any():
            inst[tree] = fill_fn(e_coords[tree] + e_off[tree], inst[tree], not_assigned_label)
        ids.copy_(torch.from_numpy(inst.astype(np.int32)))
    dist.broadcast(ids, src=src_rank, group=group)                                       # collective 3: the instance ids
    return e_coords, ids.cpu().numpy().astype(np.int64)
