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
//   lrb   : per output row 32 x u16: entry k = LDS byte offset of tap k's input row inside the unit's stage (position * 64 + swizzle;
//           own rows at positions 0..63, halo rows from position 64, absent -> the all-zero row at position 191).
//   pmask : 27-bit presence mask per (new) row.
//
// Integer / bit work on the occupancy bitmap + popcount prefix of tl_voxel.hip; no MFMA.  Everything is a pure function of the
// bitmap (deterministic, no atomics) except the ORDER of the appended units, which follows an atomic counter.
#include "tl_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int HASH = 2048;                 // per-wave hash slots (<= 64 * 26 = 1664 distinct outside rows)
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint32_t KEYMASK = 0x01FFFFFFu;  // rows < 2^25; the upper 7 bits of a slot receive the halo rank
constexpr int LIST = 128;

struct BlkGrid { int B, BX, BY, BZ; int64_t nblk; };

__device__ __forceinline__ void blk_decode(const BlkGrid& g, int64_t blk, int& b, int& bx, int& by, int& bz) {
  bz = (int)(blk % g.BZ); blk /= g.BZ;
  by = (int)(blk % g.BY); blk /= g.BY;
  bx = (int)(blk % g.BX);
  b = (int)(blk / g.BX);
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
                                                      int32_t* __restrict__ o2n, int32_t* __restrict__ perm, int32_t* __restrict__ coords_new) {
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
  while (byte) {
    const int zb = __ffs((int)byte) - 1;
    byte &= byte - 1;
    o2n[old] = newr; perm[newr] = old;
    reinterpret_cast<int4*>(coords_new)[newr] = make_int4(b, x, y, bz * 8 + zb);
    ++old; ++newr;
  }
}

struct BlkOut {
  const int32_t* o2n; const int32_t* coords_new;
  int32_t* unit; int32_t* counter;      // counter[0] = number of units (pre-set to n_chunks), counter[1] = error flag
  int32_t* halo; uint16_t* lrb; int32_t* pmask;
  int64_t n; int64_t nchunks; int64_t cap_units; int halo_max;
};

// units, halo lists, local rulebooks: one wave per chunk of 64 new rows
__global__ void __launch_bounds__(kBlock) k_blk_units(const uint64_t* __restrict__ bm, const uint32_t* __restrict__ pf, TlDims d, BlkOut p) {
  __shared__ uint32_t s_tab[kBlock / 64][HASH];
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

  // the lane's 27 neighbours as NEW row ids (-1 = absent): the rank probes of tl_voxel.hip's rulebook kernel, then o2n
  int nn[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) nn[k] = -1;
  uint32_t pm = 0;
  if (rvalid) {
    const int4 c = reinterpret_cast<const int4*>(p.coords_new)[r];
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        const int x = c.y + dx, y = c.z + dy;
        const int tap0 = (dx + 1) * 9 + (dy + 1) * 3;
        int r0 = -1, r1 = -1, r2 = -1;
        if (x >= 0 && x < d.X && y >= 0 && y < d.Y) {
          const int64_t wc = tl_col_word(d, c.x, x, y);
          const int z = c.w;
          const int64_t w = wc + (z >> 6);
          const uint64_t word = bm[w];
          const uint32_t pbase = pf[w];
          const int bit = z & 63;
          const uint64_t below = (1ull << bit) - 1;
          if (word & (1ull << bit)) r1 = (int)(pbase + __popcll(word & below));
          if (bit > 0) { if (word & (1ull << (bit - 1))) r0 = (int)(pbase + __popcll(word & (below >> 1))); }
          else if (z > 0) r0 = tl_rank_at(bm, pf, wc, z - 1);
          if (bit < 63) { if (word & (2ull << bit)) r2 = (int)(pbase + __popcll(word & ((below << 1) | 1ull))); }
          else if (z + 1 < d.Z) r2 = tl_rank_at(bm, pf, wc, z + 1);
        }
        if (r0 >= 0) { nn[tap0] = p.o2n[r0]; pm |= 1u << tap0; }
        if (r1 >= 0) { nn[tap0 + 1] = p.o2n[r1]; pm |= 2u << tap0; }
        if (r2 >= 0) { nn[tap0 + 2] = p.o2n[r2]; pm |= 4u << tap0; }
      }
    }
    p.pmask[r] = (int32_t)pm;
  }

  // the hash table is sized to the chunk: >= 2 x (outside references), 512 .. HASH slots
  int nref = 0;
#pragma unroll
  for (int k = 0; k < 27; ++k) nref += (nn[k] >= 0 && (nn[k] < (int)base || nn[k] >= (int)base + cnt)) ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) nref += __shfl_xor(nref, off);
  const int TS = nref <= 256 ? 512 : (nref <= 512 ? 1024 : HASH);
  const int tshift = nref <= 256 ? 23 : (nref <= 512 ? 22 : 21);
  auto hs = [&](uint32_t v) __attribute__((always_inline)) { return (v * 2654435761u) >> tshift; };

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
    // Insert without atomics (an LDS compare-and-swap costs ~200 clocks per wave instruction) and without one LDS round trip per tap
    // (the first version was latency-bound on ~190 dependent round trips per chunk): nine taps at a time, every pending key is read,
    // written where its slot looks empty, read back; a key is placed when its slot holds it and moves to the next slot when another
    // key won the slot.  The reads / writes of a round are independent of each other.
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      uint32_t sl[9], pend = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int v = nn[9 * g + t];
        if (inr && v >= 0 && (v < lo || v >= hi)) pend |= 1u << t;
        sl[t] = hs((uint32_t)v);
      }
      while (__any(pend != 0)) {
        uint32_t cur[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) cur[t] = ((pend >> t) & 1u) ? tab[sl[t]] : 0u;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (((pend >> t) & 1u) && cur[t] == EMPTY) tab[sl[t]] = (uint32_t)nn[9 * g + t];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
        for (int t = 0; t < 9; ++t) cur[t] = ((pend >> t) & 1u) ? tab[sl[t]] : 0u;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if ((pend >> t) & 1u) {
            if (cur[t] == (uint32_t)nn[9 * g + t]) pend &= ~(1u << t); else sl[t] = (sl[t] + 1) & (uint32_t)(TS - 1);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      }
    }
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
      const uint32_t k0 = lane < H ? list[lane] : EMPTY, k1 = lane + 64 < H ? list[lane + 64] : EMPTY;
      const uint32_t s0 = lane < H ? lslot[lane] : 0u, s1 = lane + 64 < H ? lslot[lane + 64] : 0u;
      int r0 = 0, r1 = 0;
      const int h0 = H < 64 ? H : 64;
      for (int i = 0; i < h0; ++i) {
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)k0, i);
        r0 += b < k0 ? 1 : 0; r1 += b < k1 ? 1 : 0;
      }
      for (int i = 64; i < H; ++i) {
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)k1, i - 64);
        r0 += b < k0 ? 1 : 0; r1 += b < k1 ? 1 : 0;
      }
      const int H16 = (H + 15) & ~15;
      if (lane < H) { p.halo[(int64_t)lo * 32 + r0] = (int32_t)k0; tab[s0] = k0 | ((uint32_t)r0 << 25); }
      if (lane + 64 < H) { p.halo[(int64_t)lo * 32 + r1] = (int32_t)k1; tab[s1] = k1 | ((uint32_t)r1 << 25); }
      if (lane >= H && lane < H16) p.halo[(int64_t)lo * 32 + lane] = -1;
      if (lane + 64 >= H && lane + 64 < H16) p.halo[(int64_t)lo * 32 + lane + 64] = -1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    // local rulebook rows: own rows directly, outside rows through the table, again nine taps at a time
    {
      uint32_t wds[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) wds[q] = 0;
      constexpr uint32_t ZV = (uint32_t)(191 * 64 + 3 * 16);
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        uint32_t sl[9], pend = 0;
        int pos[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int v = nn[9 * g + t];
          pos[t] = 191;
          if (inr && v >= 0) {
            if (v >= lo && v < hi) pos[t] = v - lo; else pend |= 1u << t;
          }
          sl[t] = hs((uint32_t)v);
        }
        while (__any(pend != 0)) {
          uint32_t cur[9];
#pragma unroll
          for (int t = 0; t < 9; ++t) cur[t] = ((pend >> t) & 1u) ? tab[sl[t]] : 0u;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            if ((pend >> t) & 1u) {
              if ((cur[t] & KEYMASK) == (uint32_t)nn[9 * g + t]) { pos[t] = 64 + (int)(cur[t] >> 25); pend &= ~(1u << t); }
              else sl[t] = (sl[t] + 1) & (uint32_t)(TS - 1);
            }
          }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int k = 9 * g + t;
          const uint32_t val = (uint32_t)(pos[t] * 64 + ((pos[t] >> 2) & 3) * 16);
          wds[k >> 1] |= val << ((k & 1) * 16);
        }
      }
      wds[13] |= ZV << 16; wds[14] = ZV | (ZV << 16); wds[15] = ZV | (ZV << 16);       // entries 27..31
      if (inr) {
        uint4* dst = reinterpret_cast<uint4*>(p.lrb + r * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = make_uint4(wds[4 * q], wds[4 * q + 1], wds[4 * q + 2], wds[4 * q + 3]);
      }
    }
    if (lane == 0) reinterpret_cast<int4*>(p.unit)[u] = make_int4(lo, e - a, H, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  }
}

__global__ void k_blk_init(int32_t* counter, int32_t nchunks) {
  if (threadIdx.x == 0) { counter[0] = nchunks; counter[1] = 0; }
}

}  // namespace

extern "C" {

int64_t tl_blk_ws_words(const int32_t dims[4]) {
  if (!dims) return -1;
  const int64_t nblk = (int64_t)dims[0] * ((dims[1] + 7) / 8) * ((dims[2] + 7) / 8) * ((dims[3] + 7) / 8);
  return 2 * nblk + 2 * ((nblk + 255) / 256) + 8;
}

int tl_blk_build(const uint64_t* bitmap, const uint32_t* prefix, const int32_t dims[4], int64_t n, const tl_blk* o, uint32_t* ws,
                 tl_stream_t stream) {
  if (!bitmap || !prefix || !dims || !o || !ws || n <= 0 || n >= (1 << 25)) return TL_ERR_ARG;
  if (!o->o2n || !o->perm || !o->coords_new || !o->unit || !o->counter || !o->halo || !o->lrb || !o->pmask) return TL_ERR_ARG;
  if (o->halo_max < 26 || o->halo_max > TL_BLK_HALO_MAX || o->cap_units < (n + 63) / 64) return TL_ERR_ARG;
  if (((uintptr_t)o->lrb) % 16 || ((uintptr_t)o->unit) % 16 || ((uintptr_t)o->coords_new) % 16) return TL_ERR_ARG;
  const TlDims d = tl_dims(dims);
  BlkGrid g;
  g.B = d.B; g.BX = (d.X + 7) / 8; g.BY = (d.Y + 7) / 8; g.BZ = (d.Z + 7) / 8;
  g.nblk = (int64_t)g.B * g.BX * g.BY * g.BZ;
  hipStream_t s = tl_s(stream);
  uint32_t* cnt = ws;
  uint32_t* bstart = ws + g.nblk;
  const int64_t nparts = tl_cdiv(g.nblk, kBlock);
  uint32_t* part = ws + 2 * g.nblk;
  uint32_t* part_excl = part + nparts;
  const int64_t nchunks = (n + 63) / 64;
  k_blk_init<<<1, 64, 0, s>>>(o->counter, (int32_t)nchunks);
  k_blk_count<<<(unsigned)nparts, kBlock, 0, s>>>(bitmap, d, g, cnt, part);
  k_scan_u32<<<1, 1024, 0, s>>>(part, nparts, part_excl, part_excl + nparts);
  k_blk_starts<<<(unsigned)nparts, kBlock, 0, s>>>(cnt, part_excl, g.nblk, bstart);
  k_blk_order<<<(unsigned)tl_cdiv(g.nblk, kBlock / 64), kBlock, 0, s>>>(bitmap, prefix, d, g, cnt, bstart, o->o2n, o->perm, o->coords_new);
  BlkOut p;
  p.o2n = o->o2n; p.coords_new = o->coords_new; p.unit = o->unit; p.counter = o->counter; p.halo = o->halo; p.lrb = o->lrb; p.pmask = o->pmask;
  p.n = n; p.nchunks = nchunks; p.cap_units = o->cap_units; p.halo_max = o->halo_max;
  k_blk_units<<<(unsigned)tl_cdiv(nchunks, kBlock / 64), kBlock, 0, s>>>(bitmap, prefix, d, p);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

}  // extern "C"
