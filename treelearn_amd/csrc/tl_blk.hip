// Block-local row order of a level and the staged form of its 27-tap SubM rulebook (gfx950).
//
// The output-stationary gather kernels ask the vector-memory path for 27 rows per output row whether a neighbour is present or not
// (32 B/clk/CU, an absent lane costs what a present one costs).  The block-local conv kernel (tl_conv_blk.hip) instead
// stages every input row ONCE per unit of 64 output rows in LDS and reads all taps' A fragments from there.  That needs a row order that
// is local in three dimensions and, per unit, the list of rows to stage plus a rulebook expressed in staged positions.  Built here:
//
//   order : voxels sorted by (batch, x >> 3, y >> 3, z >> 3) -- 8x8x8 blocks --, inside a block in the ascending (x, y, z) order of the
//           reference layout (a stable sort of the canonical rows by block key).  o2n / perm map canonical row <-> new row.
//   unit  : <= 64 consecutive new rows.  Chunk c = new rows [64 c, 64 c + 64) is ONE unit when the rows its taps reach outside the chunk
//           ("halo") number <= TL_BLK_HALO_MAX; otherwise the chunk is halved (recursively) until every piece fits.  The first piece of a
//           chunk is unit c, further pieces are appended behind the n_chunks regular units (rare: ~1.5 % of the chunks of a forest tile).
//   halo  : per unit the DISTINCT outside rows in ascending order (new row ids), at halo + 32 * row0, padded with -1 to a multiple of 16.
//   lrb   : per output row 9 x u32 holding 27 ten-bit entries (tap k in word k / 3, bits 10 (k % 3) ..): entry = 4 * position + swizzle, i.e.
//           entry * 16 = LDS byte offset of tap k's input row inside the unit's stage (own rows at positions 0..63, halo rows from
//           position 64, absent -> the all-zero row at position 191): 36 B per row, less than half of the 27-entry table's 108.
//   pmask : 27-bit presence mask per (new) row.
//
// Integer / bit work on the occupancy bitmap + popcount prefix of tl_voxel.hip; no MFMA.  Everything is a pure function of the
// bitmap: deterministic (the appended units are claimed through an atomic counter and then sorted by their first row).
#include "tl_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int HASH = 2048;                 // per-wave hash slots (<= 64 * 26 = 1664 distinct outside rows)
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint32_t KEYMASK = 0x01FFFFFFu;  // rows < 2^25; the upper 7 bits of a slot receive the halo rank
constexpr int LIST = 128;

// Block order: (batch, x tile, y tile, x in tile, y in tile, z) with tiles of 4 x 4 blocks -- the blocks a round of resident waves works on
// form a compact brick, so the halo rows of a unit (rows of the neighbouring blocks) are mostly in the same XCD's L2 (with z-columns of
// blocks simply ordered by (x, y), every block's x neighbours were ~1 100 units away: 18 % of the conv kernel's fetch was halo re-reads).
// Blocks of a partial tile beyond the grid are empty.
struct BlkGrid { int B, BX, BY, BZ, TX, TY; int64_t nblk; };

__device__ __forceinline__ void blk_decode(const BlkGrid& g, int64_t blk, int& b, int& bx, int& by, int& bz) {
  bz = (int)(blk % g.BZ); blk /= g.BZ;
  const int iy = (int)(blk & 3), ix = (int)((blk >> 2) & 3); blk >>= 4;
  const int ty = (int)(blk % g.TY); blk /= g.TY;
  const int tx = (int)(blk % g.TX);
  b = (int)(blk / g.TX);
  bx = tx * 4 + ix; by = ty * 4 + iy;
}

// voxels per 8x8x8 block: one thread per block, 64 bitmap bytes (one per (x, y) column of the block)
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t v, uint32_t* total) {      // kBlock threads
  __shared__ uint32_t wsum[kBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < kBlock / 64; ++w) { if (w < wid) base += wsum[w]; tot += wsum[w]; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// cnt[blk] and, per workgroup of kBlock blocks, their total (part[workgroup])
__global__ void __launch_bounds__(kBlock) k_blk_count(const uint64_t* __restrict__ bm, TlDims d, BlkGrid g, uint32_t* __restrict__ cnt,
                                                      uint32_t* __restrict__ part) {
  const int64_t blk = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  uint32_t c = 0;
  if (blk < g.nblk) {
    int b, bx, by, bz;
    blk_decode(g, blk, b, bx, by, bz);
    const int sh = (bz & 7) * 8;
    for (int xi = 0; xi < 8; ++xi) {
      const int x = bx * 8 + xi;
      if (x >= d.X) break;
      for (int yi = 0; yi < 8; ++yi) {
        const int y = by * 8 + yi;
        if (y >= d.Y) break;
        const uint64_t w = bm[tl_col_word(d, b, x, y) + (bz >> 3)];
        c += __popc((uint32_t)(w >> sh) & 0xFFu);
      }
    }
    cnt[blk] = c;
  }
  uint32_t tot;
  wg_exclusive_scan(c, &tot);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// bstart[blk] = part_excl[workgroup] + exclusive scan of cnt inside the workgroup
__global__ void __launch_bounds__(kBlock) k_blk_starts(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ part_excl, int64_t nblk,
                                                       uint32_t* __restrict__ bstart) {
  const int64_t blk = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint32_t c = blk < nblk ? cnt[blk] : 0;
  uint32_t tot;
  const uint32_t ex = wg_exclusive_scan(c, &tot);
  if (blk < nblk) bstart[blk] = part_excl[blockIdx.x] + ex;
}

// exclusive scan of n u32 values by ONE workgroup of 1024 threads: every thread owns a contiguous slice
__global__ void __launch_bounds__(1024) k_scan_u32(const uint32_t* __restrict__ in, int64_t n, uint32_t* __restrict__ out, uint32_t* __restrict__ total) {
  __shared__ uint32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t per = (n + 1023) / 1024, lo = (int64_t)tid * per, hi = lo + per < n ? lo + per : n;
  uint32_t s = 0;
  for (int64_t i = lo; i < hi; ++i) s += in[i];
  uint32_t inc = s;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < 16; ++w) { if (w < wid) base += wsum[w]; tot += wsum[w]; }
  uint32_t run = base + inc - s;
  for (int64_t i = lo; i < hi; ++i) { const uint32_t v = in[i]; out[i] = run; run += v; }
  if (tid == 0 && total) *total = tot;
}

// new row of every voxel: one wave per block, lane = (x, y) column of the block
__global__ void __launch_bounds__(kBlock) k_blk_order(const uint64_t* __restrict__ bm, const uint32_t* __restrict__ pf, TlDims d, BlkGrid g,
                                                      const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ bstart,
                                                      int32_t* __restrict__ o2n, int32_t* __restrict__ perm, int32_t* __restrict__ coords_new,
                                                      uint32_t* __restrict__ cs) {
  const int lane = threadIdx.x & 63;
  const int64_t blk = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (blk >= g.nblk) return;
  if (cnt[blk] == 0) return;                               // wave-uniform
  int b, bx, by, bz;
  blk_decode(g, blk, b, bx, by, bz);
  const int x = bx * 8 + (lane >> 3), y = by * 8 + (lane & 7);
  const bool valid = x < d.X && y < d.Y;
  const int sh = (bz & 7) * 8;
  uint64_t word = 0; int64_t w = 0;
  if (valid) { w = tl_col_word(d, b, x, y) + (bz >> 3); word = bm[w]; }
  uint32_t byte = (uint32_t)(word >> sh) & 0xFFu;
  const uint32_t pc = __popc(byte);
  uint32_t inc = pc;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)inc, off);
    if (lane >= off) inc += t;
  }
  if (!pc) return;
  int newr = (int)(bstart[blk] + inc - pc);
  int old = (int)(pf[w] + __popcll(word & ((1ull << sh) - 1ull)));
  cs[(((int64_t)b * d.X + x) * d.Y + y) * g.BZ + bz] = (uint32_t)newr;       // new row of the first voxel of this (column, z byte)
  while (byte) {
    const int zb = __ffs((int)byte) - 1;
    byte &= byte - 1;
    o2n[old] = newr; perm[newr] = old;
    reinterpret_cast<int4*>(coords_new)[newr] = make_int4(b, x, y, bz * 8 + zb);
    ++old; ++newr;
  }
}

struct BlkOut {
  const uint32_t* cs; int BZ;           // new row of the first voxel of every non-empty (column, z byte): [b][x][y][BZ]
  const int32_t* coords_new;
  int32_t* unit; int32_t* counter;      // counter[0] = number of units (pre-set to n_chunks), counter[1] = error flag
  int32_t* halo; uint32_t* lrb; int32_t* pmask; int32_t* nn_out;
  int64_t n; int64_t nchunks; int64_t cap_units; int halo_max;
};

#ifdef TL_DEV
__device__ int g_blk_abl = 0;              // developer build: k_blk_units stops after stage g_blk_abl (tools/dev_blk_units.py); results are wrong on purpose
#define BLK_ABL(stage) if (g_blk_abl == (stage)) return
#else
#define BLK_ABL(stage)
#endif

// units, halo lists, local rulebooks: one wave per chunk of 64 new rows
__global__ void __launch_bounds__(kBlock) k_blk_units(const uint64_t* __restrict__ bm, TlDims d, BlkOut p) {
  __shared__ uint32_t s_tab[kBlock / 64][HASH + 64];        // + the dummy slot idle lanes use
  __shared__ uint32_t s_list[kBlock / 64][LIST];
  __shared__ uint32_t s_lslot[kBlock / 64][LIST];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t chunk = (int64_t)blockIdx.x * (kBlock / 64) + wv;
  if (chunk >= p.nchunks) return;
  uint32_t* tab = s_tab[wv];
  uint32_t* list = s_list[wv];
  uint32_t* lslot = s_lslot[wv];
  const int64_t base = chunk * 64;
  const int cnt = (int)(p.n - base < 64 ? p.n - base : 64);
  const int64_t r = base + lane;
  const bool rvalid = lane < cnt;

  // the lane's 27 neighbours as NEW row ids (-1 = absent).  Presence comes from the occupancy bitmap; the new row of a present cell is
  // cs[column, z byte] + the set bits below it in its byte (rows of a column byte are consecutive in the block order) -- no rank
  // prefix, no canonical -> new look-up.  The loads are issued in THREE batches (coordinates; the nine column words; the column-byte
  // starts) with addresses selected instead of loads branched around: with a branch per column hipcc waits for every column's word
  // before it requests the next one, and eighteen dependent round trips at ~2 us each were the whole kernel (0.31 ms).
  int nn[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) nn[k] = -1;
  uint32_t pm = 0;
  {
    const int4 c = rvalid ? reinterpret_cast<const int4*>(p.coords_new)[r] : make_int4(0, 0, 0, 0);
    const int z = c.w;
    const bool hz0 = z > 0, hz2 = z + 1 < d.Z;
    const int bA = (hz0 ? z - 1 : z) >> 3, bB = (hz2 ? z + 1 : z) >> 3;
    const bool x0 = hz0 && ((z - 1) >> 6) != (z >> 6), x2 = hz2 && ((z + 1) >> 6) != (z >> 6);     // dz = -1 / +1 lies in the next word
    // column (x + dx, y + dy) = the lane's own column + a WAVE-UNIFORM offset (scalar registers); a column outside the grid reads the
    // lane's own column instead (always valid) and is masked afterwards
    const int64_t wc0 = rvalid ? tl_col_word(d, c.x, c.y, c.z) : 0;
    const int64_t ci0 = rvalid ? (((int64_t)c.x * d.X + c.y) * d.Y + c.z) * p.BZ : 0;
    const bool xin[3] = {rvalid && c.y > 0, rvalid, rvalid && c.y + 1 < d.X}, yin[3] = {c.z > 0, true, c.z + 1 < d.Y};
    const int zw1 = rvalid ? (z >> 6) : 0, zw0 = (rvalid && x0) ? ((z - 1) >> 6) : zw1, zw2 = (rvalid && x2) ? ((z + 1) >> 6) : zw1;
    bool inb[9];
    uint64_t w1[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int dx = q / 3 - 1, dy = q % 3 - 1;
      inb[q] = xin[q / 3] && yin[q % 3];
      w1[q] = bm[wc0 + (inb[q] ? (dx * d.Y + dy) * d.Zw : 0) + zw1];                      // (the offset is wave-uniform)
    }
    // the bytes that hold z - 1, z, z + 1 of every column, packed (bits 0-7, 8-15, 16-23); the 64-bit words go away
    uint32_t by3[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const uint32_t b1 = (uint32_t)(w1[q] >> (z & 56)) & 0xFFu;
      const uint32_t b0 = (uint32_t)(w1[q] >> ((hz0 ? z - 1 : z) & 56)) & 0xFFu, b2 = (uint32_t)(w1[q] >> ((hz2 ? z + 1 : z) & 56)) & 0xFFu;
      by3[q] = b0 | (b1 << 8) | (b2 << 16);
    }
    if (__any(x0 || x2)) {                                   // rare (z on a 64-cell boundary): dz = -1 / +1 lies in the neighbouring word
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int dx = q / 3 - 1, dy = q % 3 - 1;
        const int64_t wcq = wc0 + (inb[q] ? (dx * d.Y + dy) * d.Zw : 0);
        const uint64_t a0 = bm[wcq + zw0], a2 = bm[wcq + zw2];
        if (x0) by3[q] = (by3[q] & ~0xFFu) | ((uint32_t)(a0 >> ((z - 1) & 56)) & 0xFFu);
        if (x2) by3[q] = (by3[q] & ~0xFF0000u) | (((uint32_t)(a2 >> ((z + 1) & 56)) & 0xFFu) << 16);
      }
    }
    bool p0[9], p1[9], p2[9];
    uint32_t sA[9], sB[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int dx = q / 3 - 1, dy = q % 3 - 1;
      p0[q] = inb[q] && hz0 && ((by3[q] >> ((z - 1) & 7)) & 1u);
      p1[q] = inb[q] && ((by3[q] >> (8 + (z & 7))) & 1u);
      p2[q] = inb[q] && hz2 && ((by3[q] >> (16 + ((z + 1) & 7))) & 1u);
      // (only where a cell is present: the starts of the ~80 % absent columns would each touch a cache line of their own)
      const bool any = p0[q] || p1[q] || p2[q];
      const int64_t ciq = ci0 + (inb[q] ? (dx * d.Y + dy) * p.BZ : 0);
      sA[q] = p.cs[any ? ciq + bA : 0];
      sB[q] = p.cs[(any && bB != bA) ? ciq + bB : 0];
    }
    auto rowid = [](uint32_t byte, int zz, uint32_t start) __attribute__((always_inline)) {
      return (int)(start + __popc(byte & ((1u << (zz & 7)) - 1u)));
    };
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const uint32_t b_ = bB != bA ? sB[q] : sA[q];
      if (p0[q]) { nn[3 * q] = rowid(by3[q] & 0xFFu, z - 1, sA[q]); pm |= 1u << (3 * q); }
      if (p1[q]) { nn[3 * q + 1] = rowid((by3[q] >> 8) & 0xFFu, z, (z >> 3) == bA ? sA[q] : b_); pm |= 2u << (3 * q); }
      if (p2[q]) { nn[3 * q + 2] = rowid((by3[q] >> 16) & 0xFFu, z + 1, b_); pm |= 4u << (3 * q); }
    }
    if (rvalid) {
      p.pmask[r] = (int32_t)pm;
      if (p.nn_out) {
#pragma unroll
        for (int k = 0; k < 27; ++k) p.nn_out[(int64_t)k * p.n + r] = nn[k];
      }
    }
  }

  BLK_ABL(1);                                                               // the neighbour ids and presence masks only
  // the hash table is sized to the chunk (load factor <= 0.7 even if every present neighbour were a distinct outside row)
  int nref = __popc(pm);
  for (int off = 32; off > 0; off >>= 1) nref += __shfl_xor(nref, off);
  const int TS = nref <= 360 ? 512 : (nref <= 720 ? 1024 : HASH);
  const int tshift = nref <= 360 ? 23 : (nref <= 720 ? 22 : 21);
  // slot of every neighbour (hoisted: the halving re-uses them); tab[TS] is a dummy slot that idle lanes read and write, so the passes
  // below need no exec-mask juggling per tap (the first version spent 1 700 scalar instructions per wave on that)
  auto hs = [&](int v) __attribute__((always_inline)) { return ((uint32_t)v * 2654435761u) >> tshift; };      // (recomputed where used: 27 registers less)

  // depth-first halving of the chunk until every piece's halo fits
  int st_a[8], st_e[8];
  int sp = 1;
  st_a[0] = 0; st_e[0] = cnt;
  bool first = true;
  while (sp > 0) {
    --sp;
    const int a = st_a[sp], e = st_e[sp];
    const int lo = (int)base + a, hi = (int)base + e;                       // the piece's new rows [lo, hi)
    for (int i = lane; i < TS; i += 64) tab[i] = EMPTY;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const bool inr = lane >= a && lane < e;
    // Insert without atomics (an LDS compare-and-swap costs ~200 clocks per wave instruction): every outside neighbour is WRITTEN to
    // its home slot (equal keys agree, of different keys one wins), then read back; the few keys that lost their slot go through a
    // probing loop (write where the slot looks empty, read back, move on while another key holds it).
    uint32_t outm = 0;                                                      // bit k: tap k reaches outside the piece
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const int v = nn[k];
      const bool out = inr && v >= 0 && (v < lo || v >= hi);
      outm |= out ? (1u << k) : 0u;
      tab[out ? hs(v) : (uint32_t)TS] = (uint32_t)v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    uint32_t lost = 0;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const uint32_t cur = tab[hs(nn[k])];
      lost |= (((outm >> k) & 1u) && cur != (uint32_t)nn[k]) ? (1u << k) : 0u;
    }
    if (__any(lost != 0)) {
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        bool todo = (lost >> k) & 1u;
        if (__any(todo)) {
          uint32_t s_ = hs(nn[k]);
          while (__any(todo)) {
            if (todo) {
              s_ = (s_ + 1) & (uint32_t)(TS - 1);
              if (tab[s_] == EMPTY) tab[s_] = (uint32_t)nn[k];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            if (todo && tab[s_] == (uint32_t)nn[k]) todo = false;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          }
        }
      }
    }
    BLK_ABL(2);                                                             // + table clear, insert, verify, probing
    // the distinct outside rows, compacted in slot order: list[i] = key, lslot[i] = its slot
    int H = 0;
    for (int i0 = 0; i0 < TS; i0 += 512) {
      uint32_t sv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) sv[q] = tab[i0 + q * 64 + lane];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned long long m = __ballot(sv[q] != EMPTY);
        if (sv[q] != EMPTY) {
          const int at = H + __popcll(m & ((1ull << lane) - 1ull));
          if (at < LIST) { list[at] = sv[q]; lslot[at] = (uint32_t)(i0 + q * 64 + lane); }
        }
        H += __popcll(m);
      }
    }
    BLK_ABL(3);                                                             // + compaction of the table
    if (H > p.halo_max && e - a > 1) {
      const int mid = a + (e - a) / 2;
      st_a[sp] = mid; st_e[sp] = e; ++sp;
      st_a[sp] = a; st_e[sp] = mid; ++sp;
      continue;
    }
    // a unit: its index
    int u;
    if (first) u = (int)chunk;
    else {
      int t = 0;
      if (lane == 0) t = atomicAdd(p.counter, 1);
      u = __shfl(t, 0);
    }
    first = false;
    if (u >= p.cap_units || H > LIST - 2) {                                // cannot happen with cap_units >= n and halo_max <= 126
      if (lane == 0) atomicMax(p.counter + 1, 1);
      continue;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    // rank of every key among the keys (ascending row id = staged position - 64): lane j holds keys j and j + 64, the others come by
    // readlane; the rank goes to the halo list and, packed above the key, back into the key's slot for the rulebook pass
    {
      const uint32_t k0 = lane < H ? list[lane] : EMPTY, s0 = lane < H ? lslot[lane] : (uint32_t)TS;
      int r0 = 0;
      const int H16 = (H + 15) & ~15;
      if (H <= 64) {
        for (int i = 0; i < H; ++i) r0 += (uint32_t)__builtin_amdgcn_readlane((int)k0, i) < k0 ? 1 : 0;
        if (lane < H) p.halo[(int64_t)lo * 32 + r0] = (int32_t)k0;
        tab[s0] = k0 | ((uint32_t)r0 << 25);
        if (lane >= H && lane < H16) p.halo[(int64_t)lo * 32 + lane] = -1;
      } else {
        const uint32_t k1 = lane + 64 < H ? list[lane + 64] : EMPTY, s1 = lane + 64 < H ? lslot[lane + 64] : (uint32_t)TS;
        int r1 = 0;
        for (int i = 0; i < 64; ++i) {
          const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)k0, i);
          r0 += b < k0 ? 1 : 0; r1 += b < k1 ? 1 : 0;
        }
        for (int i = 64; i < H; ++i) {
          const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)k1, i - 64);
          r0 += b < k0 ? 1 : 0; r1 += b < k1 ? 1 : 0;
        }
        p.halo[(int64_t)lo * 32 + r0] = (int32_t)k0;
        tab[s0] = k0 | ((uint32_t)r0 << 25);
        if (lane + 64 < H) p.halo[(int64_t)lo * 32 + r1] = (int32_t)k1;
        tab[s1] = k1 | ((uint32_t)r1 << 25);
        if (lane + 64 >= H && lane + 64 < H16) p.halo[(int64_t)lo * 32 + lane + 64] = -1;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    BLK_ABL(4);                                                             // + rank sort and halo list
    // local rulebook rows: own rows directly, outside rows through the table (home slot first; the few displaced keys probe on)
    {
      uint32_t wds[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) wds[q] = 0;
      uint32_t miss = 0;
      uint32_t cur[27];
#pragma unroll
      for (int k = 0; k < 27; ++k) cur[k] = tab[hs(nn[k])];
#pragma unroll
      for (int k = 0; k < 27; ++k) miss |= (((outm >> k) & 1u) && (cur[k] & KEYMASK) != (uint32_t)nn[k]) ? (1u << k) : 0u;
      if (__any(miss != 0)) {
#pragma unroll
        for (int k = 0; k < 27; ++k) {
          bool todo = (miss >> k) & 1u;
          if (__any(todo)) {
            uint32_t s_ = hs(nn[k]);
            while (__any(todo)) {
              if (todo) {
                s_ = (s_ + 1) & (uint32_t)(TS - 1);
                const uint32_t c_ = tab[s_];
                if ((c_ & KEYMASK) == (uint32_t)nn[k]) { cur[k] = c_; todo = false; }
              }
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const int v = nn[k];
        const bool out = (outm >> k) & 1u;
        const int pos = out ? 64 + (int)(cur[k] >> 25) : ((inr && v >= 0) ? v - lo : 191);
        const uint32_t val = (uint32_t)(pos * 4 + ((pos >> 2) & 3));                    // x 16 = byte offset of the row's piece 0 in the stage
        wds[k / 3] |= val << ((k % 3) * 10);
      }
      if (inr) {
        uint32_t* dst = p.lrb + r * 9;
#pragma unroll
        for (int q = 0; q < 9; ++q) dst[q] = wds[q];
      }
    }
    if (lane == 0) reinterpret_cast<int4*>(p.unit)[u] = make_int4(lo, e - a, H, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  }
}

// The pieces appended behind the regular units arrive in the order of an atomic counter; rank them by their first row so that the unit
// array -- and with it which wave works on which unit, i.e. the order in which the training epilogue's partial sums are formed -- is the
// same from run to run.  Few entries (~1.5 % of the chunks): a rank sort through a scratch copy.
// (scratch = the unused end of the unit array itself: cap_units >= n rows were reserved, nchunks + 2 m of them are needed here)
__global__ void __launch_bounds__(kBlock) k_blk_tail_rank(int32_t* __restrict__ unit, const int32_t* __restrict__ counter, int nchunks, int64_t cap_units) {
  __shared__ int keys[kBlock];
  const int m = counter[0] - nchunks;
  if (m <= 1 || (int64_t)nchunks + 2 * (int64_t)m > cap_units) return;
  const int4* u = reinterpret_cast<const int4*>(unit) + nchunks;
  int4* tmp = reinterpret_cast<int4*>(unit) + (cap_units - m);
  for (int t0 = blockIdx.x * kBlock; t0 < m; t0 += gridDim.x * kBlock) {          // (uniform per workgroup: every thread reaches the barriers)
    const int t = t0 + threadIdx.x;
    const int4 me = t < m ? u[t] : make_int4(0, 0, 0, 0);
    int rank = 0;
    for (int j0 = 0; j0 < m; j0 += kBlock) {
      keys[threadIdx.x] = j0 + (int)threadIdx.x < m ? u[j0 + threadIdx.x].x : 0x7FFFFFFF;
      __syncthreads();
      const int lim = m - j0 < kBlock ? m - j0 : kBlock;
      for (int j = 0; j < lim; ++j) rank += keys[j] < me.x ? 1 : 0;
      __syncthreads();
    }
    if (t < m) tmp[rank] = me;
  }
}
__global__ void __launch_bounds__(kBlock) k_blk_tail_copy(int32_t* __restrict__ unit, const int32_t* __restrict__ counter, int nchunks, int64_t cap_units) {
  const int m = counter[0] - nchunks;
  if (m <= 1 || (int64_t)nchunks + 2 * (int64_t)m > cap_units) return;
  const int4* tmp = reinterpret_cast<const int4*>(unit) + (cap_units - m);
  for (int t = blockIdx.x * kBlock + threadIdx.x; t < m; t += gridDim.x * kBlock) reinterpret_cast<int4*>(unit)[nchunks + t] = tmp[t];
}

__global__ void k_blk_init(int32_t* counter, int32_t nchunks) {
  if (threadIdx.x == 0) { counter[0] = nchunks; counter[1] = 0; }
}

}  // namespace

#ifdef TL_DEV
extern "C" int tl_dev_blk_abl(int stage) { return hipMemcpyToSymbol(HIP_SYMBOL(g_blk_abl), &stage, sizeof(int)) == hipSuccess ? TL_OK : TL_ERR_LAUNCH; }
#endif

extern "C" {

int64_t tl_blk_ws_words(const int32_t dims[4]) {
  if (!dims) return -1;
  const int64_t nblk = (int64_t)dims[0] * ((dims[1] + 31) / 32) * ((dims[2] + 31) / 32) * 16 * ((dims[3] + 7) / 8);
  const int64_t ncb = (int64_t)dims[0] * dims[1] * dims[2] * ((dims[3] + 7) / 8);       // (column, z byte) starts
  return 2 * nblk + 2 * ((nblk + 255) / 256) + 8 + ncb;
}

int tl_blk_build(const uint64_t* bitmap, const uint32_t* prefix, const int32_t dims[4], int64_t n, const tl_blk* o, uint32_t* ws,
                 int phases, tl_stream_t stream) {
  if (!bitmap || !prefix || !dims || !o || !ws || n <= 0 || n >= (1 << 25) || !(phases & 3)) return TL_ERR_ARG;
  if (!o->o2n || !o->perm || !o->coords_new || !o->unit || !o->counter || !o->halo || !o->lrb || !o->pmask) return TL_ERR_ARG;
  if (o->halo_max < 26 || o->halo_max > TL_BLK_HALO_MAX || o->cap_units < (n + 63) / 64) return TL_ERR_ARG;
  if (((uintptr_t)o->unit) % 16 || ((uintptr_t)o->coords_new) % 16) return TL_ERR_ARG;
  const TlDims d = tl_dims(dims);
  BlkGrid g;
  g.B = d.B; g.BX = (d.X + 7) / 8; g.BY = (d.Y + 7) / 8; g.BZ = (d.Z + 7) / 8; g.TX = (g.BX + 3) / 4; g.TY = (g.BY + 3) / 4;
  g.nblk = (int64_t)g.B * g.TX * g.TY * 16 * g.BZ;
  hipStream_t s = tl_s(stream);
  uint32_t* cnt = ws;
  uint32_t* bstart = ws + g.nblk;
  const int64_t nparts = tl_cdiv(g.nblk, kBlock);
  uint32_t* part = ws + 2 * g.nblk;
  uint32_t* part_excl = part + nparts;
  uint32_t* cs = ws + 2 * g.nblk + 2 * nparts + 8;
  const int64_t nchunks = (n + 63) / 64;
  if (phases & 1) {
    k_blk_init<<<1, 64, 0, s>>>(o->counter, (int32_t)nchunks);
    k_blk_count<<<(unsigned)nparts, kBlock, 0, s>>>(bitmap, d, g, cnt, part);
    k_scan_u32<<<1, 1024, 0, s>>>(part, nparts, part_excl, part_excl + nparts);
    k_blk_starts<<<(unsigned)nparts, kBlock, 0, s>>>(cnt, part_excl, g.nblk, bstart);
    k_blk_order<<<(unsigned)tl_cdiv(g.nblk, kBlock / 64), kBlock, 0, s>>>(bitmap, prefix, d, g, cnt, bstart, o->o2n, o->perm, o->coords_new, cs);
  }
  if (!(phases & 2)) { TL_CHECK_LAUNCH(); return TL_OK; }
  BlkOut p;
  p.cs = cs; p.BZ = g.BZ; p.coords_new = o->coords_new; p.unit = o->unit; p.counter = o->counter; p.halo = o->halo; p.lrb = o->lrb; p.pmask = o->pmask; p.nn_out = o->nn;
  p.n = n; p.nchunks = nchunks; p.cap_units = o->cap_units; p.halo_max = o->halo_max;
  k_blk_units<<<(unsigned)tl_cdiv(nchunks, kBlock / 64), kBlock, 0, s>>>(bitmap, d, p);
  k_blk_tail_rank<<<32, kBlock, 0, s>>>(o->unit, o->counter, (int)nchunks, o->cap_units);
  k_blk_tail_copy<<<32, kBlock, 0, s>>>(o->unit, o->counter, (int)nchunks, o->cap_units);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

}  // extern "C"
