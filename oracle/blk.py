"""ORACLE (test infrastructure only): numpy restatement of the block-local row order and staged rulebook form that
treelearn_amd/csrc/tl_blk.hip builds for the level-1 convs (include/treelearn_hip.h `tl_blk`).

This is an INTERNAL layout of the build (the reference's spconv has no counterpart: it keeps voxels in hash order and gathers every tap,
tree_learn/model/blocks.py:57-70), so the restatement follows the header's definition, and its anchor to the reference is the canonical
rulebook it is derived from (oracle/voxel.py `subm_table`, pinned by the golden fixtures): decoding the staged form must give back exactly
that table, row-permuted.  Only tests/ may import this module.
"""
import numpy as np

HALO_MAX = 126
ZERO_POS = 191


def block_order(coords):
    """coords i32[n,4] (b,x,y,z) in canonical (ascending key) order -> perm (new row -> canonical row), o2n (canonical -> new):
    stable sort by (b, x >> 5, y >> 5, (x >> 3) & 3, (y >> 3) & 3, z >> 3): 8x8x8 blocks, ordered in tiles of 4 x 4 block columns."""
    c = coords.astype(np.int64)
    bx, by, bz = c[:, 1] >> 3, c[:, 2] >> 3, c[:, 3] >> 3
    key = (((((c[:, 0] * 4096 + (bx >> 2)) * 4096 + (by >> 2)) * 4 + (bx & 3)) * 4 + (by & 3)) * 8192) + bz
    perm = np.argsort(key, kind="stable")
    o2n = np.empty_like(perm)
    o2n[perm] = np.arange(len(perm))
    return perm, o2n


def entry(pos):
    """ten-bit rulebook entry of a staged position: x 16 = byte offset of the row's piece 0 (64-B rows, pieces XOR-swizzled by (pos >> 2) & 3)"""
    return pos * 4 + ((pos >> 2) & 3)


def pack(ent):
    """entries i64[n, 27] -> the rulebook rows u32[n, 9] (tap k in word k // 3 at bits 10 (k % 3))"""
    out = np.zeros((ent.shape[0], 9), np.int64)
    for k in range(27):
        out[:, k // 3] |= ent[:, k] << (10 * (k % 3))
    return out.astype(np.uint32)


def unpack(lrb):
    w = lrb.astype(np.int64)
    return np.stack([(w[:, k // 3] >> (10 * (k % 3))) & 1023 for k in range(27)], 1)


def build(coords, nbr, halo_max=HALO_MAX):
    """coords i32[n,4], nbr i32[27,n] (canonical) -> dict(perm, o2n, coords_new, pmask, nn (table in new rows, i64[27,n]),
    units = list of (row0, n_own, halo rows ascending), lrb u16[n,32]).  Units: chunk c = new rows [64c, 64c+64) is one unit if its
    distinct outside neighbours number <= halo_max, else it is halved recursively (left piece first); `units` lists the first piece
    of every chunk at index c and all further pieces behind them, ascending by first row."""
    n = coords.shape[0]
    perm, o2n = block_order(coords)
    t = nbr[:, perm].astype(np.int64)
    nn = np.where(t >= 0, o2n[np.clip(t, 0, None)], -1)                     # [27, n] in new rows
    pmask = ((nn >= 0).astype(np.int64) << np.arange(27)[:, None]).sum(0).astype(np.int32)

    def halo_of(lo, hi):
        v = nn[:, lo:hi]
        out = v[(v >= 0) & ((v < lo) | (v >= hi))]
        return np.unique(out)

    nchunks = (n + 63) // 64
    first, extra = [None] * nchunks, []
    for c in range(nchunks):
        stack = [(64 * c, min(64 * c + 64, n))]
        is_first = True
        while stack:
            lo, hi = stack.pop()
            h = halo_of(lo, hi)
            if len(h) > halo_max and hi - lo > 1:
                mid = lo + (hi - lo) // 2
                stack.append((mid, hi)); stack.append((lo, mid))
                continue
            if is_first:
                first[c] = (lo, hi - lo, h)
            else:
                extra.append((lo, hi - lo, h))
            is_first = False
    units = first + sorted(extra, key=lambda u: u[0])
    ent = np.full((n, 27), entry(ZERO_POS), dtype=np.int64)
    for lo, cnt, h in units:
        v = nn[:, lo:lo + cnt].T                                              # [cnt, 27]
        inside = (v >= lo) & (v < lo + cnt)
        pos = np.full(v.shape, ZERO_POS, dtype=np.int64)
        pos[inside] = (v - lo)[inside]
        outside = (v >= 0) & ~inside
        pos[outside] = 64 + np.searchsorted(h, v[outside])
        ent[lo:lo + cnt] = entry(pos)
    return dict(perm=perm, o2n=o2n, coords_new=coords[perm], pmask=pmask, nn=nn, units=units, lrb=pack(ent))


def build_fast(coords, nbr, halo_max=HALO_MAX):
    """The same result for large inputs: chunks are evaluated together (vectorised), only the rare over-full chunks go through the
    per-chunk recursion of `build`."""
    n = coords.shape[0]
    perm, o2n = block_order(coords)
    t = nbr[:, perm].astype(np.int64)
    nn = np.where(t >= 0, o2n[np.clip(t, 0, None)], -1)
    pmask = ((nn >= 0).astype(np.int64) << np.arange(27)[:, None]).sum(0).astype(np.int32)
    nchunks = (n + 63) // 64
    chunk = np.arange(n) // 64
    lo_r = chunk * 64
    outside = (nn >= 0) & ((nn < lo_r[None, :]) | (nn >= (lo_r + 64)[None, :]))
    pk = np.unique(np.broadcast_to(chunk[None, :], nn.shape)[outside] * n + nn[outside])      # (chunk, outside row), sorted
    hc = pk // n
    cnt = np.bincount(hc, minlength=nchunks)
    start = np.cumsum(cnt) - cnt
    big = set(np.nonzero(cnt > halo_max)[0].tolist())

    def halo_of(lo, hi):
        v = nn[:, lo:hi]
        return np.unique(v[(v >= 0) & ((v < lo) | (v >= hi))])

    first, extra = [None] * nchunks, []
    for c in range(nchunks):
        lo, hi = 64 * c, min(64 * c + 64, n)
        if c not in big:
            first[c] = (lo, hi - lo, pk[start[c]:start[c] + cnt[c]] % n)
            continue
        stack = [(lo, hi)]
        is_first = True
        while stack:
            a, e = stack.pop()
            h = halo_of(a, e)
            if len(h) > halo_max and e - a > 1:
                mid = a + (e - a) // 2
                stack.append((mid, e)); stack.append((a, mid))
                continue
            if is_first:
                first[c] = (a, e - a, h)
            else:
                extra.append((a, e - a, h))
            is_first = False
    units = first + sorted(extra, key=lambda u: u[0])
    # local rulebook, vectorised over all rows: unit of every row, then position of every (row, tap)
    u_lo = np.array([u[0] for u in units]); u_cnt = np.array([u[1] for u in units])
    order = np.argsort(u_lo, kind="stable")
    row_unit = np.repeat(order, u_cnt[order])                                   # unit index of every new row (rows ascending)
    lo_of = u_lo[row_unit]; hi_of = lo_of + u_cnt[row_unit]
    inside = (nn >= lo_of[None, :]) & (nn < hi_of[None, :])
    pos = np.full(nn.shape, ZERO_POS, dtype=np.int64)
    pos[inside] = (nn - lo_of[None, :])[inside]
    out = (nn >= 0) & ~inside
    hl = np.concatenate([u[2] for u in units]) if units else np.zeros(0, np.int64)
    hcnt = np.array([len(u[2]) for u in units]); hstart = np.cumsum(hcnt) - hcnt
    ukey = np.repeat(np.arange(len(units)), hcnt) * n + hl                         # sorted within a unit, units ascending
    q = np.broadcast_to(row_unit[None, :], nn.shape)[out] * n + nn[out]
    pos[out] = 64 + np.searchsorted(ukey, q) - hstart[np.broadcast_to(row_unit[None, :], nn.shape)[out]]
    return dict(perm=perm, o2n=o2n, coords_new=coords[perm], pmask=pmask, nn=nn, units=units, lrb=pack(entry(pos).T))


def decode(units, lrb, n):
    """units + local rulebooks -> the table in new rows i64[27, n] (what the conv kernel contracts)."""
    out = np.full((27, n), -2, dtype=np.int64)
    for lo, cnt, h in units:
        pos = unpack(lrb[lo:lo + cnt]) >> 2
        v = np.where(pos < 64, lo + pos, -1)
        hm = (pos >= 64) & (pos != ZERO_POS)
        v[hm] = h[(pos - 64)[hm]]
        out[:, lo:lo + cnt] = v.T
    return out
