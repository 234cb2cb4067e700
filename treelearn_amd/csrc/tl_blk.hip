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
__global__ void __launch_bounds__(kBlock) k_blk_count(const uint64_t* __restrict__ bm, TlDims d, BlkGrid g, uint32_t* __restrict__ cnt) {
  const int64_t blk = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (blk >= g.nblk) return;
  int b, bx, by, bz;
  blk_decode(g, blk, b, bx, by, bz);
  const int sh = (bz & 7) * 8;
  uint32_t c = 0;
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

__device__ __forceinline__ uint32_t hslot(uint32_t v) { return (v * 2654435761u) >> 21; }     // 11 bits

// units, halo lists, local rulebooks: one wave per chunk of 64 new rows
__global__ void __launch_bounds__(kBlock) k_blk_units(const uint64_t* __restrict__ bm, const uint32_t* __restrict__ pf, TlDims d, BlkOut p) {
  __shared__ uint32_t s_tab[kBlock / 64][HASH];
  __shared__ uint32_t s_list[kBlock / 64][LIST];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t chunk = (int64_t)blockIdx.x * (kBlock / 64) + wv;
  if (chunk >= p.nchunks) return;
  uint32_t* tab = s_tab[wv];
  uint32_t* list = s_list[wv];
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

  // depth-first halving of the chunk until every piece's halo fits
  int st_a[8], st_e[8];
  int sp = 1;
  st_a[0] = 0; st_e[0] = cnt;
  bool first = true;
  while (sp > 0) {
    --sp;
    const int a = st_a[sp], e = st_e[sp];
    const int lo = (int)base + a, hi = (int)base + e;                       // the piece's new rows [lo, hi)
    for (int i = lane; i < HASH; i += 64) tab[i] = EMPTY;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const bool inr = lane >= a && lane < e;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const int v = nn[k];
      if (inr && v >= 0 && (v < lo || v >= hi)) {
        uint32_t s = hslot((uint32_t)v);
        while (true) {
          const uint32_t old = atomicCAS(&tab[s], EMPTY, (uint32_t)v);
          if (old == EMPTY || old == (uint32_t)v) break;
          s = (s + 1) & (HASH - 1);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    int H = 0;
    for (int i = 0; i < HASH; i += 64) H += __popcll(__ballot(tab[i + lane] != EMPTY));
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
    // the distinct outside rows: compacted in slot order, then ranked (ascending row id = staged position - 64)
    int run = 0;
    for (int i = 0; i < HASH; i += 64) {
      const uint32_t s = tab[i + lane];
      const unsigned long long m = __ballot(s != EMPTY);
      if (s != EMPTY) list[run + __popcll(m & ((1ull << lane) - 1ull))] = s;
      run += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const int H16 = (H + 15) & ~15;
    for (int j = lane; j < H16; j += 64) {
      if (j < H) {
        const uint32_t key = list[j];
        int rank = 0;
        for (int i = 0; i < H; ++i) rank += list[i] < key ? 1 : 0;
        p.halo[(int64_t)lo * 32 + rank] = (int32_t)key;
        uint32_t s = hslot(key);                                            // leave the rank with the key for the rulebook pass
        while (tab[s] != key) s = (s + 1) & (HASH - 1);
        tab[s] = key | ((uint32_t)rank << 25);
      } else {
        p.halo[(int64_t)lo * 32 + j] = -1;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (inr) {
      uint32_t wds[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) wds[q] = 0;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        int pos = 191;
        if (k < 27) {
          const int v = nn[k];
          if (v >= 0) {
            if (v >= lo && v < hi) pos = v - lo;
            else {
              uint32_t s = hslot((uint32_t)v);
              while ((tab[s] & KEYMASK) != (uint32_t)v) s = (s + 1) & (HASH - 1);
              pos = 64 + (int)(tab[s] >> 25);
            }
          }
        }
        const uint32_t val = (uint32_t)(pos * 64 + ((pos >> 2) & 3) * 16);
        wds[k >> 1] |= val << ((k & 1) * 16);
      }
      uint4* dst = reinterpret_cast<uint4*>(p.lrb + r * 32);
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[q] = make_uint4(wds[4 * q], wds[4 * q + 1], wds[4 * q + 2], wds[4 * q + 3]);
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
  return 2 * nblk + 4;
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
  const int64_t nchunks = (n + 63) / 64;
  k_blk_init<<<1, 64, 0, s>>>(o->counter, (int32_t)nchunks);
  k_blk_count<<<(unsigned)tl_cdiv(g.nblk, kBlock), kBlock, 0, s>>>(bitmap, d, g, cnt);
  k_scan_u32<<<1, 1024, 0, s>>>(cnt, g.nblk, bstart, ws + 2 * g.nblk);
  k_blk_order<<<(unsigned)tl_cdiv(g.nblk, kBlock / 64), kBlock, 0, s>>>(bitmap, prefix, d, g, cnt, bstart, o->o2n, o->perm, o->coords_new);
  BlkOut p;
  p.o2n = o->o2n; p.coords_new = o->coords_new; p.unit = o->unit; p.counter = o->counter; p.halo = o->halo; p.lrb = o->lrb; p.pmask = o->pmask;
  p.n = n; p.nchunks = nchunks; p.cap_units = o->cap_units; p.halo_max = o->halo_max;
  k_blk_units<<<(unsigned)tl_cdiv(nchunks, kBlock / 64), kBlock, 0, s>>>(bitmap, prefix, d, p);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

}  // extern "C"
