"""Tile sharding over the GPUs of one node (the reference has no distributed code).

Tiles are independent forward passes (reference tree_learn/util/pipeline.py:83-88), so the tile loop
shards with NO data-path collective: one process per GPU, static longest-processing-time assignment
by point count, each rank runs `get_pointwise_preds` on its tiles.  The only exchange is one
variable-length all-gather of the inner-square results (4-5 % of the points), after which every rank
holds what the single-GPU loop would have produced, in the single-GPU tile order.
"""
import numpy as np
import torch
import torch.distributed as dist


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


def all_gather_rows(x: torch.Tensor, group=None):
    """All-gather tensors whose first dimension differs per rank (padded all_gather; RCCL has no
    all_gatherv).  Returns the list of per-rank tensors."""
    world = dist.get_world_size(group)
    n = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c) for c in counts]
    m = max(counts) if counts else 0
    pad = torch.zeros((m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return [b[:c] for b, c in zip(bufs, counts)]


def get_pointwise_preds_sharded(model, tiles, config, logger=None, group=None, device=None):
    """`get_pointwise_preds` over `tiles` (a sequence of batch dicts) sharded across the process
    group; every rank returns the full 8-array result in the original tile order."""
    from .pipeline import get_pointwise_preds
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = assign_tiles([t["coords"].shape[0] for t in tiles], world)[rank]
    # one pipelined pass over this rank's tiles; the per-tile row counts restore the global tile order afterwards
    res, rows = get_pointwise_preds(model, [tiles[i] for i in mine], config, logger, return_tile_rows=True)
    per_tile, off = [], 0
    for pos, n in rows:
        if n:                                            # skipped ("reach zero!!!") or empty-inner tiles contribute nothing
            per_tile.append((mine[pos], tuple(r[off:off + n] for r in res)))
        off += n
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    # header: (tile index, rows) per local tile; payload: the 8 arrays concatenated row-wise
    hdr = torch.tensor([[i, r[0].shape[0]] for i, r in per_tile], dtype=torch.int64, device=dev).reshape(-1, 2)
    hdrs = all_gather_rows(hdr, group)
    outs = []
    for a in range(8):
        parts = [torch.from_numpy(np.ascontiguousarray(r[a])) for _, r in per_tile]
        if parts:
            local = torch.cat(parts, 0)
        else:                                            # rank without tiles: empty array of the right trailing shape / dtype
            local = torch.zeros((0,), dtype=torch.float32)
        meta = torch.tensor([local.dim(), local.shape[1] if local.dim() > 1 else 0, int(local.dtype == torch.int64)], dtype=torch.int64, device=dev)
        metas = [torch.zeros_like(meta) for _ in range(world)]
        dist.all_gather(metas, meta, group=group)
        ref = max(metas, key=lambda m: int(m[0]) * 1000 + int(m[1]))      # a rank that has data defines shape/dtype
        dt = torch.int64 if any(int(m[2]) for m in metas) else torch.float32
        if local.shape[0] == 0:
            shape = (0,) if int(ref[0]) <= 1 else (0, int(ref[1]))
            local = torch.zeros(shape, dtype=dt)
        gathered = all_gather_rows(local.to(dev), group)
        # re-order tile blocks into ascending tile index
        blocks = {}
        for rk, (h, g) in enumerate(zip(hdrs, gathered)):
            off = 0
            for ti, rows in h.tolist():
                blocks[ti] = g[off:off + rows]; off += rows
        ordered = [blocks[k] for k in sorted(blocks)]
        outs.append((torch.cat(ordered, 0) if ordered else local).cpu().numpy())
    return tuple(outs)
