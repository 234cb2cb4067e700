// Voxel hashing + rulebook construction for gfx950.
//
// Integer / bit work, HBM- and L2-bound: no MFMA here.  The tile's voxel grid is held as an
// occupancy bitmap (z along the 64-bit word) plus an exclusive prefix sum of word popcounts;
// rank(cell) = prefix[word] + popc(word & below(bit)) is the voxel row in ascending (b,x,y,z)
// order.  For a 40x40 m tile at 0.1 m the bitmap is 6 MB and the prefix 3 MB: both stay in the
// 256 MB Infinity Cache (mostly in the 4 MB/XCD L2), so the 27 probes per voxel of the
// submanifold rulebook never touch HBM and need neither hashing nor sorting.
#include "tl_common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint32_t enc_f32(float f) {
  uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}

__global__ void k_init_minmax(uint32_t* __restrict__ mm, int B, int32_t* __restrict__ maxc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * 6) mm[i] = ((i % 6) < 3) ? 0xFFFFFFFFu : 0u;
  if (i < 4) maxc[i] = 0;
}

// per-batch-element min / max of xyz  (tree_learn.py:134-135).  Thread-local accumulation over the
// workgroup's contiguous slice (four independent points per iteration), then wave and workgroup reduction: 6 atomics per workgroup.
__global__ void __launch_bounds__(kBlock) k_minmax(const float* __restrict__ xyz, const int64_t* __restrict__ bid,
                                                   int64_t N_all, int B, uint32_t* __restrict__ mm) {
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  int cb = -1;
  auto flush = [&]() {
    if (cb >= 0 && cb < B)
      for (int j = 0; j < 3; ++j) { atomicMin(&mm[cb * 6 + j], lo[j]); atomicMax(&mm[cb * 6 + 3 + j], hi[j]); }
  };
  auto take = [&](int b, float x, float y, float z) {
    if (b != cb) {
      flush();
      cb = b;
      for (int j = 0; j < 3; ++j) { lo[j] = 0xFFFFFFFFu; hi[j] = 0u; }
    }
    const uint32_t e[3] = {enc_f32(x), enc_f32(y), enc_f32(z)};
    for (int j = 0; j < 3; ++j) { lo[j] = min(lo[j], e[j]); hi[j] = max(hi[j], e[j]); }
  };
  // four independent points per iteration: the loop is latency-bound otherwise (one dependent load chain per point).
  // Every workgroup walks its own CONTIGUOUS slice of the points (batch elements are contiguous ranges: dataset.py:167-226), so
  // only the few workgroups that straddle an element boundary ever flush a second element (a grid-wide stride made every thread
  // see every element: 6 same-address atomics per thread and element = 4.6 ms for a batch of two tiles).
  const int64_t per_blk = (N_all + gridDim.x - 1) / gridDim.x;
  const int64_t lo_i = (int64_t)blockIdx.x * per_blk, N = min(N_all, lo_i + per_blk);
  const int64_t stride = blockDim.x;
  int64_t i = lo_i + threadIdx.x;
  for (; i + 3 * stride < N; i += 4 * stride) {
    int b[4]; float p[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t q = i + u * stride;
      b[u] = (int)bid[q]; p[u][0] = xyz[q * 3]; p[u][1] = xyz[q * 3 + 1]; p[u][2] = xyz[q * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) take(b[u], p[u][0], p[u][1], p[u][2]);
  }
  for (; i < N; i += stride) take((int)bid[i], xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]);
  // Same-address atomics retire at ~12 ns each on this part (measured: 49 k of them = 0.5 ms), so reduce as far as possible
  // first: wave shuffle, then the workgroup through LDS -> 6 atomics per workgroup when it saw a single batch element.
  __shared__ uint32_t red[kBlock / 64][6];
  __shared__ int red_b[kBlock / 64];
  const int wid = threadIdx.x >> 6;
  const int b0 = __shfl(cb, 0);
  const bool wave_uniform = __all(cb == b0);
  if (wave_uniform) {
    for (int j = 0; j < 3; ++j)
      for (int off = 32; off > 0; off >>= 1) {
        lo[j] = min(lo[j], (uint32_t)__shfl_xor((int)lo[j], off));
        hi[j] = max(hi[j], (uint32_t)__shfl_xor((int)hi[j], off));
      }
    if ((threadIdx.x & 63) == 0) { for (int j = 0; j < 3; ++j) { red[wid][j] = lo[j]; red[wid][3 + j] = hi[j]; } red_b[wid] = cb; }
  } else {
    flush();
    if ((threadIdx.x & 63) == 0) red_b[wid] = -2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    bool same = true;
    for (int w = 0; w < kBlock / 64; ++w) same = same && red_b[w] == red_b[0] && red_b[w] >= -1;
    if (same) {
      cb = red_b[0];
      for (int j = 0; j < 3; ++j) { lo[j] = red[0][j]; hi[j] = red[0][3 + j]; }
      for (int w = 1; w < kBlock / 64; ++w)
        for (int j = 0; j < 3; ++j) { lo[j] = min(lo[j], red[w][j]); hi[j] = max(hi[j], red[w][3 + j]); }
      flush();
    } else {
      for (int w = 0; w < kBlock / 64; ++w)
        if (red_b[w] >= 0) {
          cb = red_b[w];
          for (int j = 0; j < 3; ++j) { lo[j] = red[w][j]; hi[j] = red[w][3 + j]; }
          flush();
        }
    }
  }
}

// c = floorf((p - min_b) / vs) in fp32 (IEEE sub, IEEE div): spconv PointToVoxel arithmetic
__global__ void __launch_bounds__(kBlock) k_point_coords(const float* __restrict__ xyz, const int64_t* __restrict__ bid,
                                                         int64_t N, int B, float vs, const uint32_t* __restrict__ mm,
                                                         int32_t* __restrict__ pc, int32_t* __restrict__ maxc) {
  int mx[3] = {0, 0, 0};
  int err = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)bid[i];
    int c[3] = {0, 0, 0};
    if (b < 0 || b >= B) {
      err = 1;
    } else {
      for (int j = 0; j < 3; ++j) {
        const float mn = dec_f32(mm[b * 6 + j]);
        const float q = floorf(__fdiv_rn(__fsub_rn(xyz[i * 3 + j], mn), vs));
        int v = (int)q;
        if (!(q >= 0.f) || q >= 65536.f) { err = 1; v = 0; }
        c[j] = v;
        mx[j] = max(mx[j], v);
      }
    }
    reinterpret_cast<int4*>(pc)[i] = make_int4(b, c[0], c[1], c[2]);
  }
  // block-level reduction, then ONE set of atomics per block (same-address atomics serialise at ~12 ns each)
  __shared__ int smx[kBlock / 64][4];
  for (int j = 0; j < 3; ++j) {
    for (int off = 32; off > 0; off >>= 1) mx[j] = max(mx[j], __shfl_xor(mx[j], off));
  }
  err = __any(err);
  if ((threadIdx.x & 63) == 0) { for (int j = 0; j < 3; ++j) smx[threadIdx.x >> 6][j] = mx[j]; smx[threadIdx.x >> 6][3] = err; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int e = 0;
    for (int w = 0; w < kBlock / 64; ++w) { for (int j = 0; j < 3; ++j) mx[j] = max(mx[j], smx[w][j]); e |= smx[w][3]; }
    for (int j = 0; j < 3; ++j) atomicMax(&maxc[j], mx[j]);
    if (e) atomicMax(&maxc[3], 1);
  }
}

// ---- single-tile form of the two kernels above (B = 1: the tile loop's forward).  Same arithmetic, NO same-address atomics (they retire at
// ~12 ns each: 256 x 6 of them were 18 of k_minmax's 30 us): every workgroup writes its partial minimum / maximum, k_point_coords_one folds
// the <= 256 partials at its start (one per thread), and its own per-workgroup maxima go home as they are -- the caller reads them back
// with the 16-byte extent it already waits for and folds them on the host.
__global__ void __launch_bounds__(kBlock) k_minmax_one(const float* __restrict__ xyz, const int64_t* __restrict__ bid, int64_t N_all,
                                                       uint32_t* __restrict__ parts) {
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  auto take = [&](int64_t b, float x, float y, float z) {
    if (b != 0) return;                                          // (an id outside [0, B) is reported by k_point_coords_one)
    const uint32_t e[3] = {enc_f32(x), enc_f32(y), enc_f32(z)};
    for (int j = 0; j < 3; ++j) { lo[j] = min(lo[j], e[j]); hi[j] = max(hi[j], e[j]); }
  };
  const int64_t per_blk = (N_all + gridDim.x - 1) / gridDim.x;
  const int64_t lo_i = (int64_t)blockIdx.x * per_blk, N = min(N_all, lo_i + per_blk);
  const int64_t stride = blockDim.x;
  int64_t i = lo_i + threadIdx.x;
  for (; i + 3 * stride < N; i += 4 * stride) {
    int64_t b[4]; float p[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t q = i + u * stride;
      b[u] = bid[q]; p[u][0] = xyz[q * 3]; p[u][1] = xyz[q * 3 + 1]; p[u][2] = xyz[q * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) take(b[u], p[u][0], p[u][1], p[u][2]);
  }
  for (; i < N; i += stride) take(bid[i], xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]);
  __shared__ uint32_t red[kBlock / 64][6];
  for (int j = 0; j < 3; ++j)
    for (int off = 32; off > 0; off >>= 1) {
      lo[j] = min(lo[j], (uint32_t)__shfl_xor((int)lo[j], off));
      hi[j] = max(hi[j], (uint32_t)__shfl_xor((int)hi[j], off));
    }
  if ((threadIdx.x & 63) == 0) { for (int j = 0; j < 3; ++j) { red[threadIdx.x >> 6][j] = lo[j]; red[threadIdx.x >> 6][3 + j] = hi[j]; } }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int j = threadIdx.x;
    uint32_t v = red[0][j];
    for (int w = 1; w < kBlock / 64; ++w) v = j < 3 ? min(v, red[w][j]) : max(v, red[w][j]);
    parts[blockIdx.x * 6 + j] = v;
  }
}

__global__ void __launch_bounds__(kBlock) k_point_coords_one(const float* __restrict__ xyz, const int64_t* __restrict__ bid, int64_t N, float vs,
                                                             const uint32_t* __restrict__ parts, int nparts, int32_t* __restrict__ pc,
                                                             int32_t* __restrict__ maxc_parts) {
  static_assert(kBlock >= 256, "one partial per thread");
  __shared__ uint32_t sred[kBlock / 64][3];
  __shared__ float smn[3];
  {
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if ((int)threadIdx.x < nparts) { for (int j = 0; j < 3; ++j) lo[j] = parts[threadIdx.x * 6 + j]; }
    for (int j = 0; j < 3; ++j)
      for (int off = 32; off > 0; off >>= 1) lo[j] = min(lo[j], (uint32_t)__shfl_xor((int)lo[j], off));
    if ((threadIdx.x & 63) == 0) { for (int j = 0; j < 3; ++j) sred[threadIdx.x >> 6][j] = lo[j]; }
    __syncthreads();
    if (threadIdx.x < 3) {
      uint32_t v = sred[0][threadIdx.x];
      for (int w = 1; w < kBlock / 64; ++w) v = min(v, sred[w][threadIdx.x]);
      smn[threadIdx.x] = dec_f32(v);
    }
    __syncthreads();
  }
  const float mn[3] = {smn[0], smn[1], smn[2]};
  int mx[3] = {0, 0, 0};
  int err = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = bid[i];
    int c[3] = {0, 0, 0};
    if (b != 0) {
      err = 1;
    } else {
      for (int j = 0; j < 3; ++j) {
        const float q = floorf(__fdiv_rn(__fsub_rn(xyz[i * 3 + j], mn[j]), vs));
        int v = (int)q;
        if (!(q >= 0.f) || q >= 65536.f) { err = 1; v = 0; }
        c[j] = v;
        mx[j] = max(mx[j], v);
      }
    }
    reinterpret_cast<int4*>(pc)[i] = make_int4((int)b, c[0], c[1], c[2]);
  }
  __shared__ int smx[kBlock / 64][4];
  for (int j = 0; j < 3; ++j) {
    for (int off = 32; off > 0; off >>= 1) mx[j] = max(mx[j], __shfl_xor(mx[j], off));
  }
  err = __any(err);
  if ((threadIdx.x & 63) == 0) { for (int j = 0; j < 3; ++j) smx[threadIdx.x >> 6][j] = mx[j]; smx[threadIdx.x >> 6][3] = err; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int e = 0;
    for (int w = 0; w < kBlock / 64; ++w) { for (int j = 0; j < 3; ++j) mx[j] = max(mx[j], smx[w][j]); e |= smx[w][3]; }
    reinterpret_cast<int4*>(maxc_parts)[blockIdx.x] = make_int4(mx[0], mx[1], mx[2], e);
  }
}

__global__ void __launch_bounds__(kBlock) k_set_bits(const int32_t* __restrict__ pc, int64_t N, TlDims d,
                                                     unsigned long long* __restrict__ bm) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(pc)[i];
    if (c.x < 0 || c.x >= d.B || c.y >= d.X || c.z >= d.Y || c.w >= d.Z) continue;
    const int64_t w = tl_col_word(d, c.x, c.y, c.z) + (c.w >> 6);
    const unsigned long long bit = 1ull << (c.w & 63);
    atomicOr(&bm[w], bit);       // result unused -> a non-returning L2 atomic, nothing waits on it (tiles are de-duplicated upstream: a test-first load only added latency)
  }
}

// Level-1 occupancy without atomics (tl_pyramid_build): every point stores ONE byte into a byte map (cell = 64 * word + bit; duplicates write
// the same value, so the race is benign), then one pass folds 64 bytes into each bitmap word.  The 1.9 M device-scope atomicOr of k_set_bits
// retire at ~23 G/s (0.083 ms on the config-2 tile, 0.8 ms on the 19 M-point tile); a store needs no read-modify-write at the memory side.
__global__ void __launch_bounds__(kBlock) k_set_bytes(const int32_t* __restrict__ pc, int64_t N, TlDims d, uint8_t* __restrict__ bytes) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(pc)[i];
    if (c.x < 0 || c.x >= d.B || c.y >= d.X || c.z >= d.Y || c.w >= d.Z) continue;
    const int64_t w = tl_col_word(d, c.x, c.y, c.z) + (c.w >> 6);
    bytes[w * 64 + (c.w & 63)] = 1;
  }
}
__global__ void __launch_bounds__(kBlock) k_bytes_to_bits(const uint8_t* __restrict__ bytes, int64_t nw, uint64_t* __restrict__ bm) {
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += (int64_t)gridDim.x * blockDim.x) {
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(bytes + w * 64);
    uint64_t out = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const ulonglong2 v = src[q];
      // eight bytes of 0 / 1 -> eight bits: byte j (value at bit 8 j) lands at bit 56 + j of the product
      out |= ((v.x * 0x0102040810204080ull) >> 56) << (16 * q);
      out |= ((v.y * 0x0102040810204080ull) >> 56) << (16 * q + 8);
    }
    bm[w] = out;
  }
}

__device__ __forceinline__ uint64_t pair_or_compress(uint64_t w) {  // bit i of result = w[2i] | w[2i+1]
  uint64_t t = (w | (w >> 1)) & 0x5555555555555555ull;
  t = (t | (t >> 1)) & 0x3333333333333333ull;
  t = (t | (t >> 2)) & 0x0f0f0f0f0f0f0f0full;
  t = (t | (t >> 4)) & 0x00ff00ff00ff00ffull;
  t = (t | (t >> 8)) & 0x0000ffff0000ffffull;
  t = (t | (t >> 16)) & 0x00000000ffffffffull;
  return t;
}

// word i of the coarse bitmap: OR of the 2x2x2 children, cells at or beyond out_shape dropped
__device__ __forceinline__ uint64_t down_word(const uint64_t* __restrict__ fine, const TlDims& f, int ox, int oy, int oz,
                                              const TlDims& c, int64_t i) {
  const int zw = (int)(i % c.Zw);
  int64_t r = i / c.Zw;
  const int y = (int)(r % c.Y); r /= c.Y;
  const int x = (int)(r % c.X);
  const int b = (int)(r / c.X);
  uint64_t lo = 0, hi = 0;
  if (x < ox && y < oy) {
    for (int dx = 0; dx < 2; ++dx)
      for (int dy = 0; dy < 2; ++dy) {
        const int fx = 2 * x + dx, fy = 2 * y + dy;
        if (fx >= f.X || fy >= f.Y) continue;
        const int64_t wc = tl_col_word(f, b, fx, fy);
        if (2 * zw < f.Zw) lo |= fine[wc + 2 * zw];
        if (2 * zw + 1 < f.Zw) hi |= fine[wc + 2 * zw + 1];
      }
  }
  uint64_t w = pair_or_compress(lo) | (pair_or_compress(hi) << 32);
  const int z0 = zw * 64;                       // drop cells at or beyond out_shape.z
  if (oz <= z0) w = 0;
  else if (oz < z0 + 64) w &= (1ull << (oz - z0)) - 1;
  return w;
}

__global__ void __launch_bounds__(kBlock) k_bitmap_down(const uint64_t* __restrict__ fine, TlDims f, int ox, int oy, int oz,
                                                        uint64_t* __restrict__ coarse, TlDims c) {
  const int64_t n = tl_nwords(c);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    coarse[i] = down_word(fine, f, ox, oy, oz, c, i);
}

// ---------------------------------------------------------------- popcount exclusive scan (3 passes)
constexpr int kScanItems = 8;                       // words per thread
constexpr int kScanTile = kBlock * kScanItems;      // words per block
constexpr int kScanMaxLevels = 8;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total_out) {
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
  *total_out = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(kBlock) k_scan_partials(const uint64_t* __restrict__ bm, int64_t n, uint32_t* __restrict__ part) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint32_t s = 0;
  for (int j = 0; j < kScanItems; ++j) if (base + j < n) s += __popcll(bm[base + j]);
  uint32_t tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(kBlock) k_scan_blocks(uint32_t* __restrict__ part, int64_t nb, uint32_t* __restrict__ total) {
  uint32_t carry = 0;                               // single block walks the partials
  for (int64_t base = 0; base < nb; base += kBlock) {
    const int64_t i = base + threadIdx.x;
    const uint32_t v = i < nb ? part[i] : 0;
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan(v, &tot);
    if (i < nb) part[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(kBlock) k_scan_final(const uint64_t* __restrict__ bm, int64_t n, const uint32_t* __restrict__ part,
                                                       uint32_t* __restrict__ prefix) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint32_t c[kScanItems];
  uint32_t s = 0;
  for (int j = 0; j < kScanItems; ++j) { c[j] = (base + j < n) ? __popcll(bm[base + j]) : 0; s += c[j]; }
  uint32_t tot;
  uint32_t ex = block_exclusive_scan(s, &tot) + part[blockIdx.x];
  for (int j = 0; j < kScanItems; ++j) { if (base + j < n) prefix[base + j] = ex; ex += c[j]; }
}

// The popcount scans of SEVERAL levels in one launch per pass (tl_pyramid_build: the big levels of the pyramid; their bitmaps are complete
// before the first pass starts).  Workgroup b works on level l with first[l] <= b < first[l + 1]; the partials of a level are contiguous.
struct ScanPack {
  int nl;
  int first[kScanMaxLevels + 1];
  int64_t off[kScanMaxLevels];          // word offset of the level's bitmap / prefix
  int64_t nw[kScanMaxLevels];
};
__device__ __forceinline__ int scan_level(const ScanPack& p) {
  int l = 0;
  while (l + 1 < p.nl && (int)blockIdx.x >= p.first[l + 1]) ++l;
  return l;
}
__global__ void __launch_bounds__(kBlock) k_scan_partials_multi(const uint64_t* __restrict__ bm_all, ScanPack p, uint32_t* __restrict__ part) {
  const int l = scan_level(p);
  const uint64_t* bm = bm_all + p.off[l];
  const int64_t n = p.nw[l];
  const int64_t base = (int64_t)(blockIdx.x - p.first[l]) * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint32_t s = 0;
  for (int j = 0; j < kScanItems; ++j) if (base + j < n) s += __popcll(bm[base + j]);
  uint32_t tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(kBlock) k_scan_blocks_multi(uint32_t* __restrict__ part, ScanPack p, uint32_t* __restrict__ totals) {
  for (int l = 0; l < p.nl; ++l) {                  // single block: level after level, each walking its own partials
    uint32_t carry = 0;
    const int64_t lo = p.first[l], nb = p.first[l + 1] - p.first[l];
    for (int64_t base = 0; base < nb; base += kBlock) {
      const int64_t i = base + threadIdx.x;
      const uint32_t v = i < nb ? part[lo + i] : 0;
      uint32_t tot;
      const uint32_t ex = block_exclusive_scan(v, &tot);
      if (i < nb) part[lo + i] = carry + ex;
      carry += tot;
    }
    if (threadIdx.x == 0) totals[l] = carry;
  }
}
__global__ void __launch_bounds__(kBlock) k_scan_final_multi(const uint64_t* __restrict__ bm_all, ScanPack p, const uint32_t* __restrict__ part,
                                                             uint32_t* __restrict__ prefix_all) {
  const int l = scan_level(p);
  const uint64_t* bm = bm_all + p.off[l];
  uint32_t* prefix = prefix_all + p.off[l];
  const int64_t n = p.nw[l];
  const int64_t base = (int64_t)(blockIdx.x - p.first[l]) * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint32_t c[kScanItems];
  uint32_t s = 0;
  for (int j = 0; j < kScanItems; ++j) { c[j] = (base + j < n) ? __popcll(bm[base + j]) : 0; s += c[j]; }
  uint32_t tot;
  uint32_t ex = block_exclusive_scan(s, &tot) + part[blockIdx.x];
  for (int j = 0; j < kScanItems; ++j) { if (base + j < n) prefix[base + j] = ex; ex += c[j]; }
}

// The deep levels of the pyramid are a few thousand words each: one 1024-thread workgroup builds all of them (down-sample,
// popcount scan, total) level after level instead of four launches per level.  Every thread scans the words it wrote itself;
// the next level reads its neighbours' words after the workgroup barrier.
constexpr int kPyrBlock = 1024;
constexpr int kPyrMaxLevels = 8;
constexpr int64_t kPyrSmallWords = 4096;
struct PyrSmall {
  int nl, first;                       // levels first .. first+nl-1 are built here
  TlDims d[kPyrMaxLevels + 1];         // d[0] = the level they start from, d[1+j] = level first+j
  int out[kPyrMaxLevels][3];           // out_shape of level first+j
  int64_t off[kPyrMaxLevels + 1];      // word offsets into the bitmap / prefix arrays, like d
};

__global__ void __launch_bounds__(kPyrBlock) k_pyramid_small(uint64_t* bm_all, uint32_t* __restrict__ pf_all,
                                                             uint32_t* __restrict__ counts, PyrSmall p) {
  __shared__ uint32_t wsum[kPyrBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int j = 0; j < p.nl; ++j) {
    const uint64_t* fine = bm_all + p.off[j];
    uint64_t* coarse = bm_all + p.off[j + 1];
    uint32_t* prefix = pf_all + p.off[j + 1];
    const TlDims f = p.d[j], c = p.d[j + 1];
    const int64_t n = tl_nwords(c);
    uint32_t carry = 0;
    for (int64_t base = 0; base < n; base += kPyrBlock) {
      const int64_t i = base + threadIdx.x;
      uint32_t v = 0;
      if (i < n) {
        const uint64_t w = down_word(fine, f, p.out[j][0], p.out[j][1], p.out[j][2], c, i);
        coarse[i] = w;
        v = __popcll(w);
      }
      uint32_t inc = v;
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += t;
      }
      if (lane == 63) wsum[wid] = inc;
      __syncthreads();
      uint32_t before = 0, tot = 0;
      for (int w = 0; w < kPyrBlock / 64; ++w) { if (w < wid) before += wsum[w]; tot += wsum[w]; }
      __syncthreads();
      if (i < n) prefix[i] = carry + before + inc - v;
      carry += tot;
    }
    if (threadIdx.x == 0) counts[p.first + j] = carry;
    __threadfence_block();
    __syncthreads();                    // level first+j complete before first+j+1 reads it
  }
}

// Bodies of the per-level kernels take (first item, stride) so that one launch can also cover several small levels
// (k_*_multi below: each workgroup finds its level from a table of first-workgroup indices).
__device__ __forceinline__ void expand_coords_body(const uint64_t* __restrict__ bm, const uint32_t* __restrict__ pf, const TlDims& d,
                                                   int32_t* __restrict__ coords, int64_t first, int64_t stride) {
  const int64_t n = tl_nwords(d);
  for (int64_t i = first; i < n; i += stride) {
    uint64_t w = bm[i];
    if (!w) continue;
    const int zw = (int)(i % d.Zw);
    int64_t r = i / d.Zw;
    const int y = (int)(r % d.Y); r /= d.Y;
    const int x = (int)(r % d.X);
    const int b = (int)(r / d.X);
    int64_t row = pf[i];
    while (w) {
      const int bit = __ffsll((unsigned long long)w) - 1;
      reinterpret_cast<int4*>(coords)[row++] = make_int4(b, x, y, zw * 64 + bit);
      w &= w - 1;
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_expand_coords(const uint64_t* __restrict__ bm, const uint32_t* __restrict__ pf, TlDims d,
                                                          int32_t* __restrict__ coords) {
  expand_coords_body(bm, pf, d, coords, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// o2n (optional): the level's block-local order (tl_blk.hip) -- the map then holds NEW rows
__global__ void __launch_bounds__(kBlock) k_point_rank(const int32_t* __restrict__ pc, int64_t N, const uint64_t* __restrict__ bm,
                                                       const uint32_t* __restrict__ pf, TlDims d, int64_t* __restrict__ v2p,
                                                       const int32_t* __restrict__ o2n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(pc)[i];
    int r = -1;
    if (c.x >= 0 && c.x < d.B && c.y < d.X && c.z < d.Y && c.w < d.Z) r = tl_rank_at(bm, pf, tl_col_word(d, c.x, c.y, c.z), c.w);
    if (o2n && r >= 0) r = o2n[r];
    v2p[i] = r;
  }
}

// ---------------------------------------------------------------- rulebooks
__device__ __forceinline__ void rulebook_subm_body(const int32_t* __restrict__ coords, int64_t M, const uint64_t* __restrict__ bm,
                                                   const uint32_t* __restrict__ pf, const TlDims& d, int32_t* __restrict__ nbr,
                                                   int32_t* __restrict__ compact, int64_t first, int64_t stride) {
  for (int64_t i = first; i < M; i += stride) {
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    uint32_t cmask = 0;
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
          // one word + its prefix serve all three dz taps unless z sits on a word edge
          const int64_t w = wc + (z >> 6);
          const uint64_t word = bm[w];
          const uint32_t base = pf[w];
          const int bit = z & 63;
          const uint64_t below = (1ull << bit) - 1;
          if (word & (1ull << bit)) r1 = (int)(base + __popcll(word & below));
          if (bit > 0) { if (word & (1ull << (bit - 1))) r0 = (int)(base + __popcll(word & (below >> 1))); }
          else if (z > 0) r0 = tl_rank_at(bm, pf, wc, z - 1);
          if (bit < 63) { if (word & (2ull << bit)) r2 = (int)(base + __popcll(word & ((below << 1) | 1ull))); }
          else if (z + 1 < d.Z) r2 = tl_rank_at(bm, pf, wc, z + 1);
        }
        nbr[(int64_t)(tap0 + 0) * M + i] = r0;
        nbr[(int64_t)(tap0 + 1) * M + i] = r1;
        nbr[(int64_t)(tap0 + 2) * M + i] = r2;
        if (compact) {                                       // column form (see k_table_compact): first present dz neighbour + presence bits
          compact[(int64_t)(tap0 / 3) * M + i] = r0 >= 0 ? r0 : (r1 >= 0 ? r1 : r2);
          cmask |= ((r0 >= 0 ? 1u : 0u) | (r1 >= 0 ? 2u : 0u) | (r2 >= 0 ? 4u : 0u)) << tap0;
        }
      }
    }
    if (compact) compact[(int64_t)9 * M + i] = (int32_t)cmask;
  }
}

__global__ void __launch_bounds__(kBlock) k_rulebook_subm(const int32_t* __restrict__ coords, int64_t M, const uint64_t* __restrict__ bm,
                                                          const uint32_t* __restrict__ pf, TlDims d, int32_t* __restrict__ nbr,
                                                          int32_t* __restrict__ compact) {
  rulebook_subm_body(coords, M, bm, pf, d, nbr, compact, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// fo2n (optional): block-local order of the FINE level (tl_blk.hip): its rows appear in that order in all three tables
__device__ __forceinline__ void rulebook_down_body(const int32_t* __restrict__ cc, int64_t Mc, const uint64_t* __restrict__ fbm,
                                                   const uint32_t* __restrict__ fpf, const TlDims& f, int64_t Mf, int32_t* __restrict__ child,
                                                   int32_t* __restrict__ parent, int32_t* __restrict__ inv, int64_t first, int64_t stride,
                                                   const int32_t* __restrict__ fo2n = nullptr, int32_t* __restrict__ invp = nullptr) {
  for (int64_t q = first; q < Mc; q += stride) {
    const int4 c = reinterpret_cast<const int4*>(cc)[q];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int x = 2 * c.y + (k >> 2), y = 2 * c.z + ((k >> 1) & 1), z = 2 * c.w + (k & 1);
      int r = -1;
      if (x < f.X && y < f.Y && z < f.Z) r = tl_rank_at(fbm, fpf, tl_col_word(f, c.x, x, y), z);
      if (fo2n && r >= 0) r = fo2n[r];
      child[(int64_t)k * Mc + q] = r;
      if (r >= 0) {                                   // (each of the three forms is optional: tl_level)
        if (parent) parent[r] = (int)q;
        if (inv) inv[(int64_t)k * Mf + r] = (int)q;
        if (invp) invp[r] = (int)((q << 3) | k);
      }
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_rulebook_down(const int32_t* __restrict__ cc, int64_t Mc, const uint64_t* __restrict__ fbm,
                                                          const uint32_t* __restrict__ fpf, TlDims f, int64_t Mf,
                                                          int32_t* __restrict__ child, int32_t* __restrict__ parent, int32_t* __restrict__ inv,
                                                          const int32_t* __restrict__ fo2n, int32_t* __restrict__ invp) {
  rulebook_down_body(cc, Mc, fbm, fpf, f, Mf, child, parent, inv, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, fo2n, invp);
}

// One launch over several small levels: workgroup b works on level l with first[l] <= b < first[l+1].
struct LevelPack {
  int nl;
  int first[kPyrMaxLevels + 1];
  TlDims d[kPyrMaxLevels];
  int64_t n[kPyrMaxLevels];
  const uint64_t* bm[kPyrMaxLevels];
  const uint32_t* pf[kPyrMaxLevels];
  int32_t* coords[kPyrMaxLevels];
  int32_t* nbr[kPyrMaxLevels];
  int32_t* child[kPyrMaxLevels];      // tables between level l (fine) and l+1 (coarse); the launch walks the coarse rows
  int32_t* parent[kPyrMaxLevels];
  int32_t* inv[kPyrMaxLevels];
};
__device__ __forceinline__ int pack_level(const LevelPack& p) {
  int l = 0;
  while (l + 1 < p.nl && (int)blockIdx.x >= p.first[l + 1]) ++l;
  return l;
}
__global__ void __launch_bounds__(kBlock) k_expand_coords_multi(LevelPack p) {
  const int l = pack_level(p);
  expand_coords_body(p.bm[l], p.pf[l], p.d[l], p.coords[l], (int64_t)(blockIdx.x - p.first[l]) * kBlock + threadIdx.x,
                     (int64_t)(p.first[l + 1] - p.first[l]) * kBlock);
}
__global__ void __launch_bounds__(kBlock) k_rulebook_subm_multi(LevelPack p) {
  const int l = pack_level(p);
  rulebook_subm_body(p.coords[l], p.n[l], p.bm[l], p.pf[l], p.d[l], p.nbr[l], nullptr,
                     (int64_t)(blockIdx.x - p.first[l]) * kBlock + threadIdx.x, (int64_t)(p.first[l + 1] - p.first[l]) * kBlock);
}
__global__ void __launch_bounds__(kBlock) k_rulebook_down_multi(LevelPack p) {    // here level slot l = the FINE level; coarse = slot l+1
  const int l = pack_level(p);
  rulebook_down_body(p.coords[l + 1], p.n[l + 1], p.bm[l], p.pf[l], p.d[l], p.n[l], p.child[l], p.parent[l], p.inv[l],
                     (int64_t)(blockIdx.x - p.first[l]) * kBlock + threadIdx.x, (int64_t)(p.first[l + 1] - p.first[l]) * kBlock);
}

// 27-tap SubM rulebook -> column form: for each of the 9 (dx, dy) columns the row index of its first present dz neighbour,
// plus one 27-bit presence mask.  The present neighbours of a column are CONSECUTIVE rows (rows are in ascending (b, x, y, z)
// order with z fastest), so tap k = 3c + d is base[c] + popc(mask bits 3c .. k-1): 40 B per voxel instead of 108.
__global__ void __launch_bounds__(kBlock) k_table_compact(const int32_t* __restrict__ t, int64_t n, int32_t* __restrict__ o) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
    uint32_t mask = 0;
    for (int c = 0; c < 9; ++c) {
      int base = -1;
      for (int d = 2; d >= 0; --d) {
        const int v = t[(int64_t)(3 * c + d) * n + r];
        if (v >= 0) { base = v; mask |= 1u << (3 * c + d); }
      }
      o[(int64_t)c * n + r] = base;
    }
    o[(int64_t)9 * n + r] = (int32_t)mask;
  }
}

// ---------------------------------------------------------------- optional voxel features (use_coords/use_feats)
__global__ void __launch_bounds__(kBlock) k_fill_i32(int32_t* p, int64_t n, int32_t v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void __launch_bounds__(kBlock) k_first_points(const int64_t* __restrict__ v2p, int64_t N, int P, int r, int32_t* __restrict__ sel) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = v2p[i];
    if (v < 0) continue;
    if (r > 0 && (int32_t)i <= sel[v * P + r - 1]) continue;   // already taken by an earlier rank
    atomicMin(&sel[v * P + r], (int32_t)i);
  }
}

__global__ void __launch_bounds__(kBlock) k_mean_feats(const float* __restrict__ pf, int C, int64_t M, int P, const int32_t* __restrict__ sel,
                                                       float* __restrict__ out) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < M * C; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = t / C; const int c = (int)(t % C);
    float s = 0.f; int n = 0;
    for (int r = 0; r < P; ++r) {
      const int32_t i = sel[v * P + r];
      if (i == 0x7FFFFFFF) break;
      bool allzero = true;                              // zero rows are padding (tree_learn.py:149-150)
      for (int j = 0; j < C; ++j) allzero &= (pf[(int64_t)i * C + j] == 0.f);
      if (allzero) continue;
      s += pf[(int64_t)i * C + c]; ++n;
    }
    out[t] = n ? __fdiv_rn(s, (float)n) : __uint_as_float(0x7FC00000u);
  }
}


// Voxel features as the network's input conv takes them (tree_learn.py:149-156), one launch: the mean of the first <= P points of a voxel over
// the columns (x, y, z, f_0 .. f_{F-1}) read from the two point arrays -- same arithmetic as k_mean_feats over their hstack: fp32 sums in point
// order, rows that are zero in every column skipped as padding, one correctly rounded division --, then ones for the column groups a flag
// switches off, the (feat.., x, y, z) column order and the cast to the compute dtype (round to nearest even, as tensor.to(dtype) rounds).
// One thread per voxel (C <= 8: the reference's default is 4).
template <typename T>
__global__ void __launch_bounds__(kBlock) k_voxel_feats(const float* __restrict__ xyz, const float* __restrict__ feats, int F, int64_t M, int P,
                                                        const int32_t* __restrict__ sel, int use_coords, int use_feats, T* __restrict__ out) {
  const int C = 3 + F;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < M; v += (int64_t)gridDim.x * blockDim.x) {
    float s[8];
    for (int c = 0; c < C; ++c) s[c] = 0.f;
    int n = 0;
    for (int r = 0; r < P; ++r) {
      const int32_t i = sel[v * P + r];
      if (i == 0x7FFFFFFF) break;
      float row[8];
      bool allzero = true;
      for (int c = 0; c < C; ++c) {
        row[c] = c < 3 ? xyz[(int64_t)i * 3 + c] : feats[(int64_t)i * F + (c - 3)];
        allzero &= (row[c] == 0.f);
      }
      if (allzero) continue;
      for (int c = 0; c < C; ++c) s[c] += row[c];
      ++n;
    }
    for (int c = 0; c < C; ++c) {
      float m = n ? __fdiv_rn(s[c], (float)n) : __uint_as_float(0x7FC00000u);
      if (c < 3 ? !use_coords : !use_feats) m = 1.f;
      out[v * C + (c < 3 ? F + c : c - 3)] = (T)m;
    }
  }
}

}  // namespace

extern "C" {

int tl_voxel_point_coords(const float* xyz, const int64_t* batch_ids, int64_t N, int B, float voxel_size,
                          uint32_t* ws_minmax, int32_t* pcoords, int32_t* maxc, tl_stream_t stream) {
  if (!xyz || !batch_ids || !ws_minmax || !pcoords || !maxc || N <= 0 || B <= 0 || !(voxel_size > 0.f)) return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  k_init_minmax<<<tl_cdiv(B * 6 > 4 ? B * 6 : 4, 64), 64, 0, s>>>(ws_minmax, B, maxc);
  k_minmax<<<tl_grid(N, kBlock * 4) < 256 ? tl_grid(N, kBlock * 4) : 256, kBlock, 0, s>>>(xyz, batch_ids, N, B, ws_minmax);
  k_point_coords<<<tl_grid(N, kBlock) < 384 ? tl_grid(N, kBlock) : 384, kBlock, 0, s>>>(xyz, batch_ids, N, B, voxel_size, ws_minmax, pcoords, maxc);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

// Single-tile form for tl_forward (B = 1; not part of the public header): no atomics, partial results instead.  ws: 6 * 256 words;
// maxc_parts: [TL_POINT_COORDS_MAX_PARTS][4] words, *n_parts rows of it are written (x, y, z maxima and the error flag of one workgroup each).
int tl_voxel_point_coords_one(const float* xyz, const int64_t* batch_ids, int64_t N, float voxel_size, uint32_t* ws, int32_t* pcoords,
                              int32_t* maxc_parts, int* n_parts, tl_stream_t stream) {
  if (!xyz || !batch_ids || !ws || !pcoords || !maxc_parts || !n_parts || N <= 0 || !(voxel_size > 0.f)) return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  const int g1 = (int)(tl_grid(N, kBlock * 4) < 256 ? tl_grid(N, kBlock * 4) : 256);
  const int g2 = (int)(tl_grid(N, kBlock * 2) < TL_POINT_COORDS_MAX_PARTS ? tl_grid(N, kBlock * 2) : TL_POINT_COORDS_MAX_PARTS);
  k_minmax_one<<<g1, kBlock, 0, s>>>(xyz, batch_ids, N, ws);
  k_point_coords_one<<<g2, kBlock, 0, s>>>(xyz, batch_ids, N, voxel_size, ws, g1, pcoords, maxc_parts);
  TL_CHECK_LAUNCH();
  *n_parts = g2;
  return TL_OK;
}

int tl_bitmap_from_points(const int32_t* pcoords, int64_t N, const int32_t dims[4], uint64_t* bitmap, tl_stream_t stream) {
  if (!pcoords || !dims || !bitmap || N <= 0) return TL_ERR_ARG;
  const TlDims d = tl_dims(dims);
  hipStream_t s = tl_s(stream);
  if (hipMemsetAsync(bitmap, 0, tl_nwords(d) * 8, s) != hipSuccess) return TL_ERR_LAUNCH;
  k_set_bits<<<tl_grid(N, kBlock), kBlock, 0, s>>>(pcoords, N, d, reinterpret_cast<unsigned long long*>(bitmap));
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_bitmap_down(const uint64_t* fine, const int32_t fdims[4], const int32_t out_shape[3], uint64_t* coarse,
                   const int32_t cdims[4], tl_stream_t stream) {
  if (!fine || !fdims || !out_shape || !coarse || !cdims) return TL_ERR_ARG;
  const TlDims f = tl_dims(fdims), c = tl_dims(cdims);
  if (c.B != f.B || c.X != (f.X + 1) / 2 || c.Y != (f.Y + 1) / 2 || c.Z != (f.Z + 1) / 2) return TL_ERR_ARG;
  k_bitmap_down<<<tl_grid(tl_nwords(c), kBlock), kBlock, 0, tl_s(stream)>>>(fine, f, out_shape[0], out_shape[1], out_shape[2], coarse, c);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int64_t tl_scan_ws_words(int64_t nwords) { return tl_cdiv(nwords, kScanTile) + 1; }

int tl_bitmap_scan(const uint64_t* bitmap, int64_t nwords, uint32_t* prefix, uint32_t* total, uint32_t* ws, tl_stream_t stream) {
  if (!bitmap || !prefix || !total || !ws || nwords <= 0) return TL_ERR_ARG;
  const int64_t nb = tl_cdiv(nwords, kScanTile);
  hipStream_t s = tl_s(stream);
  k_scan_partials<<<(unsigned)nb, kBlock, 0, s>>>(bitmap, nwords, ws);
  k_scan_blocks<<<1, kBlock, 0, s>>>(ws, nb, total);
  k_scan_final<<<(unsigned)nb, kBlock, 0, s>>>(bitmap, nwords, ws, prefix);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_expand_coords(const uint64_t* bitmap, const uint32_t* prefix, const int32_t dims[4], int32_t* coords, tl_stream_t stream) {
  if (!bitmap || !prefix || !dims || !coords) return TL_ERR_ARG;
  const TlDims d = tl_dims(dims);
  k_expand_coords<<<tl_grid(tl_nwords(d), kBlock), kBlock, 0, tl_s(stream)>>>(bitmap, prefix, d, coords);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_point_rank(const int32_t* pcoords, int64_t N, const uint64_t* bitmap, const uint32_t* prefix, const int32_t dims[4],
                  int64_t* v2p, tl_stream_t stream) {
  if (!pcoords || !bitmap || !prefix || !dims || !v2p || N <= 0) return TL_ERR_ARG;
  k_point_rank<<<tl_grid(N, kBlock), kBlock, 0, tl_s(stream)>>>(pcoords, N, bitmap, prefix, tl_dims(dims), v2p, nullptr);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_voxel_mean_feats(const float* pf, int C, const int64_t* v2p, int64_t N, int64_t M, int P, int32_t* ws, float* out,
                        tl_stream_t stream) {
  if (!pf || !v2p || !ws || !out || C <= 0 || C > 64 || N <= 0 || M <= 0 || P <= 0 || N > 0x7FFFFFFE) return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  k_fill_i32<<<tl_grid(M * P, kBlock), kBlock, 0, s>>>(ws, M * P, 0x7FFFFFFF);
  for (int r = 0; r < P; ++r) k_first_points<<<tl_grid(N, kBlock), kBlock, 0, s>>>(v2p, N, P, r, ws);
  k_mean_feats<<<tl_grid(M * C, kBlock), kBlock, 0, s>>>(pf, C, M, P, ws, out);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_voxel_feats(const float* xyz, const float* feats, int F, const int64_t* v2p, int64_t N, int64_t M, int P, int use_coords, int use_feats,
                   int dtype, int32_t* ws, void* out, tl_stream_t stream) {
  if (!xyz || !feats || !v2p || !ws || !out || F <= 0 || F > 5 || N <= 0 || M <= 0 || P <= 0 || N > 0x7FFFFFFE) return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  k_fill_i32<<<tl_grid(M * P, kBlock), kBlock, 0, s>>>(ws, M * P, 0x7FFFFFFF);
  for (int r = 0; r < P; ++r) k_first_points<<<tl_grid(N, kBlock), kBlock, 0, s>>>(v2p, N, P, r, ws);
  if (dtype == TL_F32) k_voxel_feats<float><<<tl_grid(M, kBlock), kBlock, 0, s>>>(xyz, feats, F, M, P, ws, use_coords, use_feats, static_cast<float*>(out));
  else if (dtype == TL_BF16) k_voxel_feats<__bf16><<<tl_grid(M, kBlock), kBlock, 0, s>>>(xyz, feats, F, M, P, ws, use_coords, use_feats, static_cast<__bf16*>(out));
  else if (dtype == TL_F16) k_voxel_feats<_Float16><<<tl_grid(M, kBlock), kBlock, 0, s>>>(xyz, feats, F, M, P, ws, use_coords, use_feats, static_cast<_Float16*>(out));
  else return TL_ERR_ARG;
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_rulebook_subm(const int32_t* coords, int64_t M, const uint64_t* bitmap, const uint32_t* prefix, const int32_t dims[4],
                     int32_t* nbr, int32_t* compact, tl_stream_t stream) {
  if (!coords || !bitmap || !prefix || !dims || !nbr || M <= 0) return TL_ERR_ARG;
  k_rulebook_subm<<<tl_grid(M, kBlock), kBlock, 0, tl_s(stream)>>>(coords, M, bitmap, prefix, tl_dims(dims), nbr, compact);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_rulebook_down(const int32_t* ccoords, int64_t Mc, const uint64_t* fbitmap, const uint32_t* fprefix, const int32_t fdims[4],
                     int64_t Mf, int32_t* child, int32_t* parent, int32_t* inv, tl_stream_t stream) {
  if (!ccoords || !fbitmap || !fprefix || !fdims || !child || !parent || !inv || Mc <= 0 || Mf <= 0) return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  if (hipMemsetAsync(parent, 0xFF, Mf * 4, s) != hipSuccess) return TL_ERR_LAUNCH;
  if (hipMemsetAsync(inv, 0xFF, Mf * 8 * 4, s) != hipSuccess) return TL_ERR_LAUNCH;
  k_rulebook_down<<<tl_grid(Mc, kBlock), kBlock, 0, s>>>(ccoords, Mc, fbitmap, fprefix, tl_dims(fdims), Mf, child, parent, inv, nullptr, nullptr);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

// ---- whole-pyramid entry points: the same kernels, enqueued from one call each (the per-level calls above leave the GPU
// waiting on the host between the many tiny launches of the deep levels), the deep levels batched into shared launches.
int64_t tl_pyramid_ws_words(const int32_t dims0[4], int num_levels, int64_t* level_word_offsets) {
  if (!dims0 || num_levels < 1 || num_levels > kPyrMaxLevels) return -1;
  int32_t d[4] = {dims0[0], dims0[1], dims0[2], dims0[3]};
  int64_t off = 0, ws = 0;
  for (int l = 0; l < num_levels; ++l) {
    const int64_t nw = tl_nwords(tl_dims(d));
    if (level_word_offsets) level_word_offsets[l] = off;
    off += nw;
    ws += tl_scan_ws_words(nw);                          // one partial per scan tile of EVERY level: the big levels are scanned in one launch per pass
    for (int j = 1; j < 4; ++j) d[j] = (d[j] + 1) / 2;
  }
  if (level_word_offsets) level_word_offsets[num_levels] = off;
  {                                                       // + the byte map of level 1 (64 B per bitmap word), behind the scan partials
    int32_t d0[4] = {dims0[0], dims0[1], dims0[2], dims0[3]};
    ws += 4 + 16 * tl_nwords(tl_dims(d0));               // (+ 4 words: the map starts at the next 16-byte boundary)
  }
  return ws;
}

int tl_pyramid_build(const int32_t* pcoords, int64_t N, const int32_t dims0[4], const int32_t shape0[3], int num_levels,
                     uint64_t* bitmaps, uint32_t* prefixes, uint32_t* counts, uint32_t* ws, tl_stream_t stream) {
  if (!pcoords || !dims0 || !shape0 || !bitmaps || !prefixes || !counts || !ws || N <= 0 || num_levels < 1 || num_levels > kPyrMaxLevels)
    return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  TlDims d[kPyrMaxLevels];
  int out[kPyrMaxLevels][3];
  int64_t off[kPyrMaxLevels + 1];
  int32_t dd[4] = {dims0[0], dims0[1], dims0[2], dims0[3]};
  int sh[3] = {shape0[0], shape0[1], shape0[2]};
  off[0] = 0;
  for (int l = 0; l < num_levels; ++l) {
    d[l] = tl_dims(dd);
    for (int j = 0; j < 3; ++j) out[l][j] = sh[j];
    off[l + 1] = off[l] + tl_nwords(d[l]);
    for (int j = 1; j < 4; ++j) dd[j] = (dd[j] + 1) / 2;
    for (int j = 0; j < 3; ++j) sh[j] /= 2;
  }
  {
    int64_t scan_ws = 0;
    for (int l2 = 0; l2 < num_levels; ++l2) scan_ws += tl_scan_ws_words(tl_nwords(d[l2]));
    uint8_t* bytes = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ws + scan_ws) + 15) & ~uintptr_t(15));
    const int64_t nw0 = tl_nwords(d[0]);
    if (hipMemsetAsync(bytes, 0, nw0 * 64, s) != hipSuccess) return TL_ERR_LAUNCH;
    k_set_bytes<<<tl_grid(N, kBlock), kBlock, 0, s>>>(pcoords, N, d[0], bytes);
    k_bytes_to_bits<<<tl_grid(nw0, kBlock), kBlock, 0, s>>>(bytes, nw0, bitmaps);
  }
  // the big levels: every bitmap first (one down-sampling launch per level), then ONE three-pass popcount scan over all of them (eleven small
  // dependent launches were five microseconds each on the path to the second read-back; the per-level entry points above keep the plain form)
  int l = 0;
  ScanPack sp;
  sp.first[0] = 0;
  for (; l < num_levels; ++l) {
    const int64_t nw = tl_nwords(d[l]);
    if (l > 0 && nw <= kPyrSmallWords) break;           // the rest goes into one workgroup
    if (l > 0)
      k_bitmap_down<<<tl_grid(nw, kBlock), kBlock, 0, s>>>(bitmaps + off[l - 1], d[l - 1], out[l][0], out[l][1], out[l][2], bitmaps + off[l], d[l]);
    sp.off[l] = off[l]; sp.nw[l] = nw; sp.first[l + 1] = sp.first[l] + (int)tl_cdiv(nw, kScanTile);
  }
  sp.nl = l;
  k_scan_partials_multi<<<(unsigned)sp.first[l], kBlock, 0, s>>>(bitmaps, sp, ws);
  k_scan_blocks_multi<<<1, kBlock, 0, s>>>(ws, sp, counts);
  k_scan_final_multi<<<(unsigned)sp.first[l], kBlock, 0, s>>>(bitmaps, sp, ws, prefixes);
  if (l < num_levels) {
    PyrSmall p;
    p.nl = num_levels - l; p.first = l;
    p.d[0] = d[l - 1]; p.off[0] = off[l - 1];
    for (int j = 0; j < p.nl; ++j) {
      p.d[j + 1] = d[l + j]; p.off[j + 1] = off[l + j];
      for (int k = 0; k < 3; ++k) p.out[j][k] = out[l + j][k];
    }
    k_pyramid_small<<<1, kPyrBlock, 0, s>>>(bitmaps, prefixes, counts, p);
  }
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_rulebooks_build(const tl_level* lv, int num_levels, int32_t* minus_one, int64_t minus_one_words,
                       const int32_t* pcoords, int64_t N, int64_t* v2p, tl_stream_t stream) {
  if (!lv || num_levels < 1 || num_levels > kPyrMaxLevels) return TL_ERR_ARG;
  for (int l = 0; l < num_levels; ++l) {
    if (!lv[l].bitmap || !lv[l].prefix || lv[l].n <= 0) return TL_ERR_ARG;
    if ((!lv[l].coords || !lv[l].nbr) && !lv[l].o2n) return TL_ERR_ARG;      // only a level in block-local order may go without them
    if (lv[l].nbr && !lv[l].coords) return TL_ERR_ARG;
    if (l > 0 && !lv[l].coords) return TL_ERR_ARG;                            // the down tables walk the coarse level's coordinates
    if (l + 1 < num_levels && (!lv[l].child || (!lv[l].inv && !lv[l].inv_packed))) return TL_ERR_ARG;     // (parent is optional, inv in at least one form)
    if (lv[l].inv_packed && lv[l].n >= (1ll << 28)) return TL_ERR_ARG;
  }
  hipStream_t s = tl_s(stream);
  // parent / inv default to -1: one fill when the caller carved them out of one block, else one per array
  auto inside = [&](const int32_t* q, int64_t words) { return minus_one && q >= minus_one && q + words <= minus_one + minus_one_words; };
  if (minus_one && minus_one_words > 0 && hipMemsetAsync(minus_one, 0xFF, minus_one_words * 4, s) != hipSuccess) return TL_ERR_LAUNCH;
  for (int l = 0; l + 1 < num_levels; ++l) {
    if (lv[l].parent && !inside(lv[l].parent, lv[l].n) && hipMemsetAsync(lv[l].parent, 0xFF, lv[l].n * 4, s) != hipSuccess) return TL_ERR_LAUNCH;
    if (lv[l].inv && !inside(lv[l].inv, lv[l].n * 8) && hipMemsetAsync(lv[l].inv, 0xFF, lv[l].n * 32, s) != hipSuccess) return TL_ERR_LAUNCH;
    if (lv[l].inv_packed && !inside(lv[l].inv_packed, lv[l].n) && hipMemsetAsync(lv[l].inv_packed, 0xFF, lv[l].n * 4, s) != hipSuccess) return TL_ERR_LAUNCH;
  }
  int small = num_levels;                               // levels small .. L-1 share launches
  for (int l = num_levels - 1; l >= 1; --l) {
    if (tl_nwords(tl_dims(lv[l].dims)) <= kPyrSmallWords && lv[l].n <= 64 * kPyrSmallWords && !lv[l].compact && !lv[l].o2n && !lv[l - 1].o2n &&
        !lv[l].inv_packed && !lv[l - 1].inv_packed) small = l; else break;
  }
  if (num_levels - small < 2) small = num_levels;
  for (int l = 0; l < small; ++l) {
    const TlDims d = tl_dims(lv[l].dims);
    if (lv[l].coords) k_expand_coords<<<tl_grid(tl_nwords(d), kBlock), kBlock, 0, s>>>(lv[l].bitmap, lv[l].prefix, d, lv[l].coords);
    if (lv[l].nbr) k_rulebook_subm<<<tl_grid(lv[l].n, kBlock), kBlock, 0, s>>>(lv[l].coords, lv[l].n, lv[l].bitmap, lv[l].prefix, d, lv[l].nbr, lv[l].compact);
  }
  LevelPack p;
  p.nl = num_levels - small;
  if (p.nl > 0) {
    for (int j = 0; j < p.nl; ++j) {
      const tl_level& v = lv[small + j];
      p.d[j] = tl_dims(v.dims); p.n[j] = v.n; p.bm[j] = v.bitmap; p.pf[j] = v.prefix; p.coords[j] = v.coords; p.nbr[j] = v.nbr;
      p.child[j] = v.child; p.parent[j] = v.parent; p.inv[j] = v.inv;
    }
    p.first[0] = 0;
    for (int j = 0; j < p.nl; ++j) p.first[j + 1] = p.first[j] + (int)tl_cdiv(tl_nwords(p.d[j]), kBlock);
    k_expand_coords_multi<<<p.first[p.nl], kBlock, 0, s>>>(p);
    for (int j = 0; j < p.nl; ++j) p.first[j + 1] = p.first[j] + (int)tl_cdiv(p.n[j], kBlock);
    k_rulebook_subm_multi<<<p.first[p.nl], kBlock, 0, s>>>(p);
  }
  for (int l = 0; l + 1 < num_levels && l < small; ++l) {   // fine level l big (or the last big one): its own launch
    const tl_level &f = lv[l], &c = lv[l + 1];
    k_rulebook_down<<<tl_grid(c.n, kBlock), kBlock, 0, s>>>(c.coords, c.n, f.bitmap, f.prefix, tl_dims(f.dims), f.n, f.child, f.parent, f.inv, f.o2n, f.inv_packed);
  }
  if (p.nl > 1) {                                           // fine levels small .. L-2: workgroups over the coarse rows
    p.nl -= 1;                                              // slots 0 .. nl-2 are fine levels; slot j+1 is read as the coarse one
    for (int j = 0; j < p.nl; ++j) p.first[j + 1] = p.first[j] + (int)tl_cdiv(p.n[j + 1], kBlock);
    k_rulebook_down_multi<<<p.first[p.nl], kBlock, 0, s>>>(p);
  }
  if (v2p) {
    if (!pcoords || N <= 0) return TL_ERR_ARG;
    k_point_rank<<<tl_grid(N, kBlock), kBlock, 0, s>>>(pcoords, N, lv[0].bitmap, lv[0].prefix, tl_dims(lv[0].dims), v2p, lv[0].o2n);
  }
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_rulebook_compact(const int32_t* table, int64_t n, int32_t* compact, tl_stream_t stream) {
  if (!table || !compact || n <= 0) return TL_ERR_ARG;
  k_table_compact<<<tl_grid(n, kBlock), kBlock, 0, tl_s(stream)>>>(table, n, compact);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

}  // extern "C"
