"""Tile geometry: voxel hashing + every rulebook of the U-Net, built on the GPU by the HIP library.

Host-side orchestration of include/treelearn_hip.h's voxel/rulebook entry points.  Replaces, for one
batch of tiles, the PointToVoxel call and `.tolist()` syncs of `voxelize`
(reference tree_learn/model/tree_learn.py:129-167) and spconv's lazy per-`indice_key` indice
generation (`subm1..7`, `spconv1..6`; blocks.py:57-70,104-123).  Two host syncs per batch in total:
the grid extent (3 ints) and the per-level voxel counts (L ints).
"""
import ctypes
import os
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _hip


COMPACT_MIN_ROWS = 65536       # levels from this size on also get the column form of their rulebook (40 instead of 108 B per voxel)
BLK_MIN_ROWS = 16384        # below this the level-1 convs run on the small-level kernel anyway (tl_conv_fwd's small_rows)
BLK_MAX_ROWS = (1 << 25) - 64
BLK_HALO_MAX = 126
TRACE = None                # developer: a list -> build_geometry appends (phase, time.perf_counter()) marks (tools/dev_fwd_host.py)


def _mark(name):
    if TRACE is not None:
        import time
        TRACE.append((name, time.perf_counter()))


class BlockedRulebook:
    """The block-local form of a level's 27-tap SubM rulebook (include/treelearn_hip.h `tl_blk`, csrc/tl_blk.hip): the level's rows are
    re-ordered by 8x8x8 block, blocks in tiles of 4 x 4 columns (`o2n` / `perm`: canonical row <-> new row), cut into units of <= 64 rows with their halo lists and local
    rulebooks.  Stands where the canonical table `nbr` stands (Level.nbr) when a geometry is built with `blocked=True`: every tensor
    that holds or is indexed by rows of the level -- features, `child` entries, `inv` / `parent`, `v2p` -- is then in the NEW order."""
    K = 27

    def __init__(self, n, o2n, perm, coords_new, unit, counter, halo, lrb, pmask, nn_table=None):
        self.n, self.o2n, self.perm, self.coords_new = n, o2n, perm, coords_new
        self.unit, self.counter, self.halo, self.lrb, self.pmask = unit, counter, halo, lrb, pmask
        self.nn_table = nn_table            # optional i32[27, n]: the plain table in the new row order (training: weight gradients, other widths)
        self.shape = (27, n)

    def count_pairs(self):
        """(output row, tap) pairs present -- what `(table >= 0).sum()` is for the canonical table."""
        m = self.pmask.to(torch.int64) & 0x7FFFFFF
        c = torch.zeros_like(m)
        for k in range(27):
            c += (m >> k) & 1
        return int(c.sum())

    def to_table(self):
        """The canonical-form table i32[27, n] in the NEW row order, decoded from units / halo lists / local rulebooks (tests, tools)."""
        n = self.n
        nu = int(self.counter[0])
        unit = self.unit[:nu].long()
        row0, nown = unit[:, 0], unit[:, 1]
        uid = torch.repeat_interleave(torch.arange(nu, device=unit.device), nown)          # unit of every row, in unit order
        first = torch.cumsum(nown, 0) - nown
        rows = row0[uid] + (torch.arange(uid.numel(), device=unit.device) - first[uid])
        w = self.lrb.long()[rows]                                                            # [rows, 9] words of 3 ten-bit entries
        ent = torch.stack([(w[:, k // 3] >> (10 * (k % 3))) & 1023 for k in range(27)], 1)
        pos = ent >> 2                                                                       # staged position of every (row, tap)
        r0 = row0[uid][:, None]
        halo = self.halo.long()
        hidx = (r0 * 32 + (pos - 64)).clamp(0, halo.numel() - 1)
        val = torch.where(pos < 64, r0 + pos, torch.where(pos == 191, torch.full_like(pos, -1), halo[hidx]))
        out = torch.full((n, 27), -2, dtype=torch.int64, device=unit.device)
        out[rows] = val
        return out.t().contiguous().int()

    def tensors(self):
        return [t for t in (self.o2n, self.perm, self.coords_new, self.unit, self.counter, self.halo, self.lrb, self.pmask, self.nn_table) if t is not None]


@dataclass
class Level:
    n: int                      # active voxels
    dims: tuple                 # bitmap grid (B, X, Y, Z)
    shape: tuple                # spconv spatial_shape at this level (drives the out-of-range drop)
    bitmap: torch.Tensor        # u64 words (int64 storage)
    prefix: torch.Tensor        # u32 (int32 storage)
    coords: torch.Tensor = None     # i32[n,4] (b,x,y,z), ascending key
    nbr: torch.Tensor = None        # i32[27,n]   subm{l}; a BlockedRulebook when the level is in block-local order
    nbr_ref: torch.Tensor = None    # blocked level only, on request: the canonical table (rows in canonical order)
    child: torch.Tensor = None      # i32[8,n_next]  spconv{l} (down)
    parent: torch.Tensor = None     # i32[n]
    inv: torch.Tensor = None        # i32[8,n]    spconv{l} (inverse)

    def row_coords(self):
        """(b, x, y, z) of the level's rows in the order its feature matrices use (the block-local order when the level is blocked)."""
        return self.nbr.coords_new if isinstance(self.nbr, BlockedRulebook) else self.coords


@dataclass
class TileGeometry:
    levels: List[Level]
    v2p: torch.Tensor            # i64[N]
    n_points: int
    batch_size: int
    pcoords: torch.Tensor = None
    _backing: list = field(default_factory=list)         # pooled buffers the per-level views alias
    blocked: bool = False        # level 1 lives in the block-local order (levels[0].nbr is a BlockedRulebook; v2p holds new rows)

    def tensors(self):
        """Every device tensor of this geometry (for Tensor.record_stream when built on a side stream)."""
        out = [self.v2p] + list(self._backing)
        if self.pcoords is not None:
            out.append(self.pcoords)
        for lv in self.levels:
            out += [t for t in (lv.coords, lv.nbr, lv.nbr_ref, lv.child, lv.parent, lv.inv) if isinstance(t, torch.Tensor)]
        return out


_SIDE = {}


def _side_stream(dev):
    """One side stream per (device, current stream): a forward on stream A must not share its helper stream with a forward on stream B."""
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = torch.cuda.Stream(device=dev)
    return s


def _nwords(d):
    return d[0] * d[1] * d[2] * ((d[3] + 63) // 64)


def level_shapes(shape1, num_levels):
    """spatial_shape per level: shape_{l+1} = shape_l // 2; raises the reference's skip-tile error
    (util/pipeline.py:91-97 looks for "reach zero!!!") when a dim collapses."""
    shapes = [tuple(int(s) for s in shape1)]
    for _ in range(num_levels - 1):
        nxt = tuple(s // 2 for s in shapes[-1])
        if min(nxt) <= 0:
            raise ValueError(f"sparse conv output spatial shape {list(nxt)} reach zero!!! (input shape {list(shapes[-1])})")
        shapes.append(nxt)
    return shapes


def _build_per_level(L, st, dev, pcoords, N, batch_size, extent, shapes, num_levels):
    """The same pyramid through the per-level entry points (one call per kernel group and level; `TL_GEOM=per_level`, kept for
    A/B runs and as the reference for tests of the two-call form)."""
    # 2. occupancy bitmaps + popcount prefix sums for every level (no host involvement)
    dims = [(batch_size,) + extent]
    for _ in range(num_levels - 1):
        d = dims[-1]
        dims.append((d[0], (d[1] + 1) // 2, (d[2] + 1) // 2, (d[3] + 1) // 2))
    nw = [_nwords(d) for d in dims]
    bm_all = torch.empty(sum(nw), dtype=torch.int64, device=dev)
    pf_all = torch.empty(sum(nw), dtype=torch.int32, device=dev)
    counts = torch.empty(num_levels, dtype=torch.int32, device=dev)
    scan_ws = torch.empty(int(L.tl_scan_ws_words(max(nw))), dtype=torch.int32, device=dev)
    levels = []
    off = 0
    for li in range(num_levels):
        bm = bm_all[off:off + nw[li]]; pf = pf_all[off:off + nw[li]]; off += nw[li]
        if li == 0:
            _hip.check(L.tl_bitmap_from_points(_hip.ptr(pcoords), N, _hip.dims4(dims[0]), _hip.ptr(bm), st), "tl_bitmap_from_points")
        else:
            _hip.check(L.tl_bitmap_down(_hip.ptr(levels[-1].bitmap), _hip.dims4(dims[li - 1]), _hip.dims3(shapes[li]),
                                        _hip.ptr(bm), _hip.dims4(dims[li]), st), "tl_bitmap_down")
        _hip.check(L.tl_bitmap_scan(_hip.ptr(bm), nw[li], _hip.ptr(pf), _hip.ptr(counts[li:li + 1]), _hip.ptr(scan_ws), st), "tl_bitmap_scan")
        levels.append(Level(n=0, dims=dims[li], shape=shapes[li], bitmap=bm, prefix=pf))
    ns = counts.tolist()                                   # host sync #2
    for lv, n in zip(levels, ns):
        lv.n = int(n)
        if lv.n <= 0:
            raise ValueError("sparse conv produced an empty level: output spatial shape reach zero!!!")

    # 3. coords, v2p, rulebooks
    for li, lv in enumerate(levels):
        lv.coords = torch.empty((lv.n, 4), dtype=torch.int32, device=dev)
        _hip.check(L.tl_expand_coords(_hip.ptr(lv.bitmap), _hip.ptr(lv.prefix), _hip.dims4(lv.dims), _hip.ptr(lv.coords), st), "tl_expand_coords")
        lv.nbr = torch.empty((27, lv.n), dtype=torch.int32, device=dev)
        # big levels also get the column form of their rulebook (40 instead of 108 B/voxel) for the kernels that read it; it rides on
        # the table tensor as an attribute
        ct = torch.empty((10, lv.n), dtype=torch.int32, device=dev) if lv.n >= COMPACT_MIN_ROWS else None
        _hip.check(L.tl_rulebook_subm(_hip.ptr(lv.coords), lv.n, _hip.ptr(lv.bitmap), _hip.ptr(lv.prefix), _hip.dims4(lv.dims),
                                      _hip.ptr(lv.nbr), _hip.ptr(ct), st), "tl_rulebook_subm")
        if ct is not None:
            lv.nbr._tl_compact = ct
    for li in range(num_levels - 1):
        f, c = levels[li], levels[li + 1]
        f.child = torch.empty((8, c.n), dtype=torch.int32, device=dev)
        f.parent = torch.empty(f.n, dtype=torch.int32, device=dev)
        f.inv = torch.empty((8, f.n), dtype=torch.int32, device=dev)
        _hip.check(L.tl_rulebook_down(_hip.ptr(c.coords), c.n, _hip.ptr(f.bitmap), _hip.ptr(f.prefix), _hip.dims4(f.dims), f.n,
                                      _hip.ptr(f.child), _hip.ptr(f.parent), _hip.ptr(f.inv), st), "tl_rulebook_down")
    v2p = torch.empty(N, dtype=torch.int64, device=dev)
    _hip.check(L.tl_point_rank(_hip.ptr(pcoords), N, _hip.ptr(levels[0].bitmap), _hip.ptr(levels[0].prefix), _hip.dims4(levels[0].dims),
                               _hip.ptr(v2p), st), "tl_point_rank")
    return TileGeometry(levels=levels, v2p=v2p, n_points=N, batch_size=batch_size, pcoords=pcoords, _backing=[bm_all, pf_all])


def build_geometry(coords: torch.Tensor, batch_ids: torch.Tensor, batch_size: int, voxel_size: float,
                   num_levels: int, spatial_shape: Optional[List[int]] = None, need_inverse: bool = True,
                   blocked: bool = False, ref_table: bool = False, blk_min_rows: int = None, nn_table: bool = False) -> TileGeometry:
    """`blocked`: put level 1 into the block-local row order (BlockedRulebook) when it has BLK_MIN_ROWS..BLK_MAX_ROWS voxels -- the form
    the inference engine's level-1 convs run on; the canonical level-1 table is then only built on request (`ref_table`).  Default:
    every level in the canonical order of SURVEY.md Appendix F (what the bit-exact rulebook tests pin).  `nn_table`: a blocked level
    also gets its rulebook as a plain table in the NEW row order (BlockedRulebook.nn_table: the training path's weight gradients and the
    convs the staged-unit kernel does not serve read it)."""
    L = _hip.lib()
    st = _hip.stream()
    _hip.require_cuda(coords, "coords"); _hip.require_cuda(batch_ids, "batch_ids")
    if coords.dtype != torch.float32 or batch_ids.dtype != torch.int64:
        raise TypeError("coords must be float32 [N,3] and batch_ids int64 [N]")
    N = coords.shape[0]
    dev = coords.device
    if N == 0:
        raise ValueError("empty tile")

    # 1. per-point voxel coordinates (fp32 floor((p - min_b)/vs)) + grid extent
    pcoords = torch.empty((N, 4), dtype=torch.int32, device=dev)
    maxc = torch.empty(4, dtype=torch.int32, device=dev)
    ws_mm = torch.empty(batch_size * 6, dtype=torch.int32, device=dev)
    _hip.check(L.tl_voxel_point_coords(_hip.ptr(coords), _hip.ptr(batch_ids), N, batch_size, float(voxel_size),
                                       _hip.ptr(ws_mm), _hip.ptr(pcoords), _hip.ptr(maxc), st), "tl_voxel_point_coords")
    _mark("enqueue point coords")
    mx = maxc.tolist()                                     # host sync #1
    _mark("host sync #1 (extent) returned")
    if mx[3]:
        raise ValueError("voxelize: batch id out of range or voxel coordinate outside [0, 65536)")
    extent = (mx[0] + 1, mx[1] + 1, mx[2] + 1)
    shape1 = tuple(spatial_shape) if spatial_shape is not None else extent      # tree_learn.py:86-87,165
    if any(e > s for e, s in zip(extent, shape1)):
        raise ValueError(f"tile extent {extent} voxels exceeds spatial_shape {shape1}")
    shapes = level_shapes(shape1, num_levels)

    if os.environ.get("TL_GEOM") == "per_level" and not blocked:
        return _build_per_level(L, st, dev, pcoords, N, batch_size, extent, shapes, num_levels)

    # 2. occupancy bitmaps + popcount prefix sums for every level: one call, one backing allocation
    dims = [(batch_size,) + extent]
    for _ in range(num_levels - 1):
        d = dims[-1]
        dims.append((d[0], (d[1] + 1) // 2, (d[2] + 1) // 2, (d[3] + 1) // 2))
    nw = [_nwords(d) for d in dims]
    tw = sum(nw)
    n_ws = int(L.tl_pyramid_ws_words(_hip.dims4(dims[0]), num_levels, None))
    # [bitmaps u64 tw | prefixes u32 tw | counts u32 8 | scan scratch]
    pyr = torch.empty(3 * tw + 8 + n_ws, dtype=torch.int32, device=dev)
    p0 = pyr.data_ptr()
    _hip.check(L.tl_pyramid_build(_hip.ptr(pcoords), N, _hip.dims4(dims[0]), _hip.dims3(shapes[0]), num_levels,
                                  p0, p0 + 8 * tw, p0 + 12 * tw, p0 + 12 * tw + 32, st), "tl_pyramid_build")
    _mark("enqueue pyramid")
    ns = pyr[3 * tw:3 * tw + num_levels].tolist()          # host sync #2
    _mark("host sync #2 (level counts) returned")
    bm_all = pyr[:2 * tw].view(torch.int64)
    pf_all = pyr[2 * tw:3 * tw]
    levels = []
    off = 0
    for li in range(num_levels):
        n = int(ns[li])
        if n <= 0:
            raise ValueError("sparse conv produced an empty level: output spatial shape reach zero!!!")
        levels.append(Level(n=n, dims=dims[li], shape=shapes[li], bitmap=bm_all[off:off + nw[li]], prefix=pf_all[off:off + nw[li]]))
        off += nw[li]

    # 3. coords, v2p, rulebooks: one backing allocation carved into the per-level arrays (64-word aligned), one call; the
    #    tensor views are made after the launches are queued
    # level 1 (and, with the opt-in window conv kernel, every big level) also gets the column form of its rulebook (40 instead of
    # 108 B/voxel); it rides on the table tensor as an attribute
    n1 = levels[0].n
    blocked = bool(blocked) and (BLK_MIN_ROWS if blk_min_rows is None else blk_min_rows) <= n1 <= BLK_MAX_ROWS
    want_ct = lambda lv: lv.n >= COMPACT_MIN_ROWS     # noqa: E731
    al = lambda w: (w + 63) & ~63                           # noqa: E731
    cur = 0
    def take(words):
        nonlocal cur
        o = cur; cur += al(words); return o
    o_v2p = take(2 * N)
    lay = []
    for li, lv in enumerate(levels):
        skip = blocked and li == 0 and not ref_table        # a blocked level needs neither canonical coordinates nor the canonical table
        lay.append(dict(coords=None if skip else take(4 * lv.n), nbr=None if skip else take(27 * lv.n),
                        ct=take(10 * lv.n) if (want_ct(lv) and not skip) else None,
                        child=take(8 * levels[li + 1].n) if li + 1 < num_levels else None))
    if blocked:
        # the halo lists are indexed by the unit's first row (32 slots per row: a one-row unit can have 26 outside neighbours), the
        # unit array can hold one unit per row: worst cases that never occur, reserved but not touched
        nblk_ws = int(L.tl_blk_ws_words(_hip.dims4(dims[0])))
        bl = dict(o2n=take(n1), perm=take(n1), cnew=take(4 * n1), unit=take(4 * n1), counter=take(64), halo=take(32 * n1), lrb=take(9 * n1),
                  pmask=take(n1), ws=take(nblk_ws), nn=take(27 * n1) if nn_table else None)
    o_m1 = cur                                              # parent / inv of all levels: contiguous, one fill with -1
    for li, lv in enumerate(levels[:-1]):
        lay[li]["parent"] = take(lv.n); lay[li]["inv"] = take(8 * lv.n)
    back = torch.empty(cur, dtype=torch.int32, device=dev)
    b0 = back.data_ptr()
    arr = (_hip.Level * num_levels)()
    for li, lv in enumerate(levels):
        a, y = arr[li], lay[li]
        a.dims[:] = lv.dims; a.n = lv.n
        a.bitmap = p0 + 8 * (sum(nw[:li])); a.prefix = p0 + 8 * tw + 4 * (sum(nw[:li]))
        a.coords = b0 + 4 * y["coords"] if y["coords"] is not None else None
        a.nbr = b0 + 4 * y["nbr"] if y["nbr"] is not None else None
        a.compact = b0 + 4 * y["ct"] if y["ct"] is not None else None
        a.o2n = None
        if li + 1 < num_levels:
            a.child = b0 + 4 * y["child"]; a.parent = b0 + 4 * y["parent"]; a.inv = b0 + 4 * y["inv"]
    o_m1_end = cur
    if blocked:
        bk = _hip.Blk()
        bk.o2n = b0 + 4 * bl["o2n"]; bk.perm = b0 + 4 * bl["perm"]; bk.coords_new = b0 + 4 * bl["cnew"]; bk.unit = b0 + 4 * bl["unit"]
        bk.counter = b0 + 4 * bl["counter"]; bk.halo = b0 + 4 * bl["halo"]; bk.lrb = b0 + 4 * bl["lrb"]; bk.pmask = b0 + 4 * bl["pmask"]
        bk.cap_units = n1; bk.halo_max = BLK_HALO_MAX; bk.reserved = 0
        bk.nn = b0 + 4 * bl["nn"] if bl["nn"] is not None else None
        _hip.check(L.tl_blk_build(arr[0].bitmap, arr[0].prefix, _hip.dims4(levels[0].dims), n1, ctypes.byref(bk), b0 + 4 * bl["ws"], 1, st), "tl_blk_build")
        arr[0].o2n = bk.o2n
        # The unit builder (instruction-bound, ~0.27 ms) runs on a side stream beside the rulebook kernels of the other levels -- when the
        # forward is on the DEFAULT stream (a lone forward: -0.15 ms).  Callers that keep several tiles in flight on their own streams
        # (util/pipeline.get_pointwise_preds, bench.py) already fill those gaps, and one more stream per tile in flight oversubscribes
        # the hardware queues (measured: 7.23 -> 7.67 ms per tile with three tiles in flight).  TL_BLK_SIDE=0 / 1 forces either.
        main = torch.cuda.current_stream()
        want_side = os.environ.get("TL_BLK_SIDE")
        use_side = (main == torch.cuda.default_stream(dev)) if want_side is None else want_side != "0"
        side = _side_stream(dev) if use_side else main
        side.wait_stream(main)
        with torch.cuda.stream(side):
            _hip.check(L.tl_blk_build(arr[0].bitmap, arr[0].prefix, _hip.dims4(levels[0].dims), n1, ctypes.byref(bk), b0 + 4 * bl["ws"], 2, _hip.stream()), "tl_blk_build")
    _hip.check(L.tl_rulebooks_build(arr, num_levels, b0 + 4 * o_m1, o_m1_end - o_m1, _hip.ptr(pcoords), N, b0 + 4 * o_v2p, st),
               "tl_rulebooks_build")
    if blocked and side is not main:
        # (no record_stream: the main stream waits for the side stream right here, before anything that could free or re-use the buffers
        # is enqueued on it; recording would make the caching allocator hold every geometry block until an event query, i.e. allocate anew)
        main.wait_stream(side)
    _mark("layout + enqueue blk / rulebooks")
    v2p = back[o_v2p:o_v2p + 2 * N].view(torch.int64)
    for li, lv in enumerate(levels):
        y = lay[li]
        if y["coords"] is not None:
            lv.coords = back[y["coords"]:y["coords"] + 4 * lv.n].view(lv.n, 4)
        if y["nbr"] is not None:
            lv.nbr = back[y["nbr"]:y["nbr"] + 27 * lv.n].view(27, lv.n)
        if y["ct"] is not None:
            lv.nbr._tl_compact = back[y["ct"]:y["ct"] + 10 * lv.n].view(10, lv.n)
        if li + 1 < num_levels:
            nc = levels[li + 1].n
            lv.child = back[y["child"]:y["child"] + 8 * nc].view(8, nc)
            lv.parent = back[y["parent"]:y["parent"] + lv.n]
            lv.inv = back[y["inv"]:y["inv"] + 8 * lv.n].view(8, lv.n)
    if blocked:
        v = lambda k, words: back[bl[k]:bl[k] + words]       # noqa: E731
        lv = levels[0]
        lv.nbr_ref = lv.nbr
        lv.nbr = BlockedRulebook(n1, v("o2n", n1), v("perm", n1), v("cnew", 4 * n1).view(n1, 4), v("unit", 4 * n1).view(n1, 4), v("counter", 64),
                                 v("halo", 32 * n1), v("lrb", 9 * n1).view(n1, 9), v("pmask", n1),
                                 nn_table=v("nn", 27 * n1).view(27, n1) if bl["nn"] is not None else None)
    return TileGeometry(levels=levels, v2p=v2p, n_points=N, batch_size=batch_size, pcoords=pcoords, _backing=[pyr, back], blocked=blocked)


def voxel_mean_feats(point_feats: torch.Tensor, geom: TileGeometry, max_points: int) -> torch.Tensor:
    """Mean of the first <= P points per voxel in input order (tree_learn.py:149-151); [M,C] in the
    (x,y,z,feat..) column order of `point_feats`."""
    L = _hip.lib()
    _hip.require_cuda(point_feats, "point_feats")
    M = geom.levels[0].n
    C = point_feats.shape[1]
    ws = torch.empty(M * max_points, dtype=torch.int32, device=point_feats.device)
    out = torch.empty((M, C), dtype=torch.float32, device=point_feats.device)
    _hip.check(L.tl_voxel_mean_feats(_hip.ptr(point_feats), C, _hip.ptr(geom.v2p), geom.n_points, M, max_points,
                                     _hip.ptr(ws), _hip.ptr(out), _hip.stream()), "tl_voxel_mean_feats")
    return out
