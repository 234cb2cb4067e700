// Internal declarations shared by the conv translation units.
#pragma once
#include "tl_common.h"
#include <hip/hip_bf16.h>
#include "tl_half.h"

#ifdef TL_F16_BUILD       // the float16 build of the conv units: same sources, IEEE-half conversions (tl_half.h), launchers with an _f16 suffix
#define tl_launch_conv_direct tl_launch_conv_direct_f16
#define tl_launch_conv_ones27 tl_launch_conv_ones27_f16
#define tl_launch_conv_stream tl_launch_conv_stream_f16
#define tl_stream_set_rb tl_stream_set_rb_f16
#define tl_launch_conv_streamq tl_launch_conv_streamq_f16
#define tl_launch_conv_small tl_launch_conv_small_f16
#define tl_launch_conv_tinycin tl_launch_conv_tinycin_f16
#define tl_launch_conv_bf16 tl_launch_conv_bf16_f16
#define tl_launch_conv_blk tl_launch_conv_blk_f16
#define tl_launch_conv_up tl_launch_conv_up_f16
#define g_small_mode g_small_mode_f16
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct ConvP {
  const void* in; int64_t in_ld;
  const void* w;
  const void* w_frag;   // optional fragment-order copy of w (tl_pack_weight_frag) or nullptr
  const void* w_x3;     // optional split-bf16 copy of fp32 weights (tl_pack_weight_x3) or nullptr: permits the bf16x3 contraction
  const int32_t* table;
  const int32_t* ctab;  // optional column form of a 27-tap table (tl_rulebook_compact) or nullptr
  int64_t n_out, n_in;
  int K, Cin, Cout;
  const float* in_scale; const float* in_shift; int in_relu, out_relu;
  const void* res; int64_t res_ld;
  const float* out_scale; const float* out_shift;
  void* out; int64_t out_ld;
  // optional extra views of the same result y = acc (+ residual), each with its own BN affine / ReLU:
  void* out2; int64_t out2_ld; const float* out2_scale; const float* out2_shift; int out2_relu;
  void* out3; int64_t out3_ld; const float* out3_scale; const float* out3_shift; int out3_relu;
  int nblk;
  // training-mode epilogue reductions (tl_conv_args.epi_mode): per-workgroup fp64 partial sums [workgroup][2][Cout] of
  //   TL_EPI_STATS : y and y^2 (y = acc + residual as stored) -- the batch statistics of the BatchNorm that consumes this conv's output
  //   TL_EPI_BN_BWD: the result is dy of relu(bn(x)); views get g = dy * [bn(x) > 0], sums are g and g * xhat (dbeta, dgamma)
  int epi_mode;
  double* red_part;
  int32_t* red_nparts;   // HOST out (may be NULL): partial rows the launched kernel writes (= its grid size)
  const void* bn_x; int64_t bn_x_ld;
  const float* bn_mean; const float* bn_rstd; const float* bn_scale; const float* bn_shift; int bn_relu;
  int one_hot; // every output row has at most one valid table entry (inverse conv)
  // block-local form of a 27-tap rulebook (tl_blk_build; tl_conv_args.blk_*), nullptr if absent
  const int32_t* blk_unit; const int32_t* blk_counter; const int32_t* blk_halo; const uint32_t* blk_lrb; const int32_t* blk_pmask;
  int dbg;     // developer ablation bits (tl_set_tuning "dbg"): 1 no A loads, 2 no B loads, 4 no MFMA, 8 no stores
};

static __device__ __forceinline__ float ld_elem(const float* p) { return *p; }
static __device__ __forceinline__ float ld_elem(const __hip_bfloat16* p) { return h16_lo((uint32_t)*reinterpret_cast<const uint16_t*>(p)); }   // "the 16-bit type" of this build
static __device__ __forceinline__ void st_elem(float* p, float v) { *p = v; }
static __device__ __forceinline__ float ld_elem(const _Float16* p) { return (float)*p; }
static __device__ __forceinline__ void st_elem(_Float16* p, float v) { *p = (_Float16)v; }
static __device__ __forceinline__ void st_elem(__hip_bfloat16* p, float v) { *reinterpret_cast<uint16_t*>(p) = (uint16_t)(h16_pack2(v, 0.f) & 0xFFFFu); }

// XCD-aware tile order: block b runs on XCD b % 8 (observed, speed only); give each XCD a contiguous
// range of tiles so neighbouring tiles -- which gather overlapping input rows -- share one L2.
static __device__ __forceinline__ int xcd_tile(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// (names kept from the bf16-only days: in the -DTL_F16_BUILD compilation of a unit these pack / unpack IEEE halves)
static __device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) { return h16_pack2(lo, hi); }
static __device__ __forceinline__ float bf16_lo(uint32_t u) { return h16_lo(u); }
static __device__ __forceinline__ float bf16_hi(uint32_t u) { return h16_hi(u); }

// ---- epilogue helpers: write y (fp32) through one output view
template <bool BF16>
static __device__ __forceinline__ void epi_store1(void* base, int64_t ld, const float* sc, const float* sh, int relu, int64_t row, int j, float y) {
  if (sc) y = fmaf(y, sc[j], sh[j]);
  if (relu) y = fmaxf(y, 0.f);
  if constexpr (BF16) ((uint16_t*)base)[row * ld + j] = (uint16_t)(pack_bf16x2(y, 0.f) & 0xFFFFu);
  else ((float*)base)[row * ld + j] = y;
}
// 8 consecutive channels c0..c0+7 (c0 % 8 == 0), 16-B (bf16) / 2x16-B (f32) vector stores
template <bool BF16>
static __device__ __forceinline__ void epi_store8(void* base, int64_t ld, const float* sc, const float* sh, int relu, int64_t row, int c0, const float (&y)[8]) {
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = y[q];
  if (sc) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(sc + c0), a1 = *reinterpret_cast<const f32x4*>(sc + c0 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sh + c0), b1 = *reinterpret_cast<const f32x4*>(sh + c0 + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = fmaf(v[q], a0[q], b0[q]); v[q + 4] = fmaf(v[q + 4], a1[q], b1[q]); }
  }
  if (relu) {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
  }
  if constexpr (BF16) {
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = pack_bf16x2(v[2 * q], v[2 * q + 1]);
    *reinterpret_cast<u32x4*>((uint16_t*)base + row * ld + c0) = o;
  } else {
    f32x4* d = reinterpret_cast<f32x4*>((float*)base + row * ld + c0);
    d[0] = f32x4{v[0], v[1], v[2], v[3]}; d[1] = f32x4{v[4], v[5], v[6], v[7]};
  }
}

// ---- dtype-generic MFMA step and residual helpers (bf16: one 32x32x16 MFMA per 16-B fragment pair;
// fp32: the 16 B are 4 channels -> four exact-fp32 32x32x2 MFMAs)
template <bool BF16>
static __device__ __forceinline__ void mma16(f32x16& acc, const u32x4& a, const u32x4& b) {
  if constexpr (BF16) {
    acc = h16_mfma(a, b, acc);
  } else {
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
  }
}
// ---- split-bf16 ("bf16x3") contraction of fp32 operands.  x = hi + lo + O(2^-16 |x|) with hi = the upper half of x's bits (activations; bf16(x) for the weights), lo =
// bf16(x - hi); a . b ~= alo.bhi + ahi.blo + ahi.bhi on the bf16 matrix cores with fp32 accumulation: three 32x32x16 MFMAs per
// 16 channels where the exact fp32 path issues eight 32x32x2 ones -- 3/16 of the matrix time at ~2^-15 relative error per product.
// Fragment convention (shared by the direct and the stream kernel): lane half fh of a 32-channel unit holds, for J = 0, 1, the fp32
// channels 16 J + 4 fh + {0..3} (16-B piece j = 2 J) and 16 J + 8 + 4 fh + {0..3} (piece j = 2 J + 1); tl_pack_weight_x3 stores the
// weights' hi parts of those eight channels as ONE 16-B piece at slot 2 J + fh of the unit's 128 B and the lo parts at slot 4 + 2 J + fh.
static __device__ __forceinline__ uint32_t x3_pack2(float a, float b) {
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  const v2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
static __device__ __forceinline__ void x3_split8(const u32x4& a0, const u32x4& a1, u32x4& hi, u32x4& lo) {
  // hi = the upper 16 bits of x (truncation: ONE v_perm_b32 per pair of elements), lo = bf16(x - hi) with x - hi exact in fp32
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t u0 = a0[2 * q], u1 = a0[2 * q + 1], v0 = a1[2 * q], v1 = a1[2 * q + 1];
    hi[q] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    hi[2 + q] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    lo[q] = x3_pack2(__uint_as_float(u0) - __uint_as_float(u0 & 0xFFFF0000u), __uint_as_float(u1) - __uint_as_float(u1 & 0xFFFF0000u));
    lo[2 + q] = x3_pack2(__uint_as_float(v0) - __uint_as_float(v0 & 0xFFFF0000u), __uint_as_float(v1) - __uint_as_float(v1 & 0xFFFF0000u));
  }
}
static __device__ __forceinline__ void mma16_x3(f32x16& acc, const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
}

template <bool BF16>
static __device__ __forceinline__ void res_add8(const void* res, int64_t elem, float (&v)[8]) {   // v += res[elem .. elem+7]
  if constexpr (BF16) {
    const u32x4 rv = *reinterpret_cast<const u32x4*>((const uint16_t*)res + elem);
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] += bf16_lo(rv[q]); v[2 * q + 1] += bf16_hi(rv[q]); }
  } else {
    const f32x4 r0 = *reinterpret_cast<const f32x4*>((const float*)res + elem), r1 = *reinterpret_cast<const f32x4*>((const float*)res + elem + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] += r0[q]; v[q + 4] += r1[q]; }
  }
}
// the shared row-vector epilogue: v[8] = acc (+ residual) of channels c0..c0+7 of row `row` -> up to three views
template <bool BF16>
static __device__ __forceinline__ void epi_views8(const ConvP& p, int64_t row, int c0, float (&v)[8]) {
  if (p.res) res_add8<BF16>(p.res, row * p.res_ld + c0, v);
  epi_store8<BF16>(p.out, p.out_ld, p.out_scale, p.out_shift, p.out_relu, row, c0, v);
  if (p.out2) epi_store8<BF16>(p.out2, p.out2_ld, p.out2_scale, p.out2_shift, p.out2_relu, row, c0, v);
  if (p.out3) epi_store8<BF16>(p.out3, p.out3_ld, p.out3_scale, p.out3_shift, p.out3_relu, row, c0, v);
}

// ---- training-mode epilogue (epi_mode != 0).  The kernels that support it keep the fp32 accumulators of a 32-row block in a wave-private
// LDS tile for the row-vector stage anyway; the reductions reuse that tile: the row stage writes the quantity to be summed back to
// its own place, then every lane adds ONE column over 16 rows (lane (fi, fh): column fi, rows 16 fh ..) and the two halves are combined.
static __device__ __forceinline__ float round_to(float v, bool bf16) { return bf16 ? bf16_lo(pack_bf16x2(v, 0.f)) : v; }

// row stage: v = acc of channels c0..c0+7 of `row` -> views written; on return v = the first summand (y, resp. g), q1 = the second
// (y^2 is formed in the column pass, so q1 is only set for TL_EPI_BN_BWD: g * xhat)
template <bool BF16>
static __device__ __forceinline__ void epi_views8_red(const ConvP& p, int64_t row, int c0, float (&v)[8], float (&q1)[8]) {
  if (p.res) res_add8<BF16>(p.res, row * p.res_ld + c0, v);
  if (p.epi_mode == TL_EPI_BN_BWD) {
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    res_add8<BF16>(p.bn_x, row * p.bn_x_ld + c0, x);
    const f32x4 m0 = *reinterpret_cast<const f32x4*>(p.bn_mean + c0), m1 = *reinterpret_cast<const f32x4*>(p.bn_mean + c0 + 4);
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(p.bn_rstd + c0), r1 = *reinterpret_cast<const f32x4*>(p.bn_rstd + c0 + 4);
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(p.bn_scale + c0), a1 = *reinterpret_cast<const f32x4*>(p.bn_scale + c0 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bn_shift + c0), b1 = *reinterpret_cast<const f32x4*>(p.bn_shift + c0 + 4);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float mu = q < 4 ? m0[q & 3] : m1[q & 3], rs = q < 4 ? r0[q & 3] : r1[q & 3], sc = q < 4 ? a0[q & 3] : a1[q & 3], sh = q < 4 ? b0[q & 3] : b1[q & 3];
      const bool keep = !p.bn_relu || fmaf(x[q], sc, sh) > 0.f;
      v[q] = keep ? round_to(v[q], BF16) : 0.f;
      q1[q] = v[q] * ((x[q] - mu) * rs);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = round_to(v[q], BF16);
  }
  epi_store8<BF16>(p.out, p.out_ld, p.out_scale, p.out_shift, p.out_relu, row, c0, v);
  if (p.out2) epi_store8<BF16>(p.out2, p.out2_ld, p.out2_scale, p.out2_shift, p.out2_relu, row, c0, v);
  if (p.out3) epi_store8<BF16>(p.out3, p.out3_ld, p.out3_scale, p.out3_shift, p.out3_relu, row, c0, v);
}

// second summand of TL_EPI_BN_BWD recomputed from the stored g (the persistent kernels do this instead of keeping q1 of every row
// vector in registers across the first column pass): q1 = g * xhat, x re-read (a cache hit)
template <bool BF16>
static __device__ __forceinline__ void epi_bnb_q1(const ConvP& p, int64_t row, int c0, const float (&g)[8], float (&q1)[8]) {
  float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  res_add8<BF16>(p.bn_x, row * p.bn_x_ld + c0, x);
  const f32x4 m0 = *reinterpret_cast<const f32x4*>(p.bn_mean + c0), m1 = *reinterpret_cast<const f32x4*>(p.bn_mean + c0 + 4);
  const f32x4 r0 = *reinterpret_cast<const f32x4*>(p.bn_rstd + c0), r1 = *reinterpret_cast<const f32x4*>(p.bn_rstd + c0 + 4);
#pragma unroll
  for (int q = 0; q < 8; ++q) q1[q] = g[q] * ((x[q] - (q < 4 ? m0[q & 3] : m1[q & 3])) * (q < 4 ? r0[q & 3] : r1[q & 3]));
}

// column pass over a [32][EP] fp32 tile: sum (and sum of squares) of column `col`; every lane returns the total of its column
template <bool SQ>
static __device__ __forceinline__ void tile_colsum(const float* ew, int EP, int col, int fh, float& s, float& ss) {
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float t = ew[(16 * fh + j) * EP + col];
    a += t;
    if constexpr (SQ) b = fmaf(t, t, b);
  }
  s = a + __shfl_xor(a, 32, 64);
  if constexpr (SQ) ss = b + __shfl_xor(b, 32, 64);
}

// the stream kernels' row stage of one 32 x 32 block (tile ew [32][EP], rows row0.., columns col0..) with the optional reductions:
// red0 / red1 += this block's column sums (lane's column = col0 + (lane & 31))
template <bool BF16, int EP>
static __device__ __forceinline__ void epi_block32(const ConvP& p, float* ew, int lane, int64_t row0, int col0, float& red0, float& red1) {
  const int fi = lane & 31, fh = lane >> 5;
  if (p.epi_mode == TL_EPI_NONE) {
#pragma unroll
    for (int e0 = 0; e0 < 2; ++e0) {
      const int e = lane + e0 * 64;                        // 32 rows x 4 vectors of 8 channels
      const int rr = e >> 2, cvv = e & 3;
      const int64_t orow = row0 + rr;
      if (orow < p.n_out) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8), v1 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        epi_views8<BF16>(p, orow, col0 + cvv * 8, v);
      }
    }
    return;
  }
  float q1s[2][8];
#pragma unroll
  for (int e0 = 0; e0 < 2; ++e0) {
    const int e = lane + e0 * 64;
    const int rr = e >> 2, cvv = e & 3;
    const int64_t orow = row0 + rr;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8), v1 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8 + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
    for (int q = 0; q < 8; ++q) q1s[e0][q] = 0.f;
    if (orow < p.n_out) epi_views8_red<BF16>(p, orow, col0 + cvv * 8, v, q1s[e0]);
    else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = 0.f;
    }
    *reinterpret_cast<f32x4*>(ew + rr * EP + cvv * 8) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(ew + rr * EP + cvv * 8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float s = 0.f, ss = 0.f;
  if (p.epi_mode == TL_EPI_STATS) tile_colsum<true>(ew, EP, fi, fh, s, ss);
  else {
    float dummy;
    tile_colsum<false>(ew, EP, fi, fh, s, dummy);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e0 = 0; e0 < 2; ++e0) {
      const int e = lane + e0 * 64;
      const int rr = e >> 2, cvv = e & 3;
      *reinterpret_cast<f32x4*>(ew + rr * EP + cvv * 8) = f32x4{q1s[e0][0], q1s[e0][1], q1s[e0][2], q1s[e0][3]};
      *reinterpret_cast<f32x4*>(ew + rr * EP + cvv * 8 + 4) = f32x4{q1s[e0][4], q1s[e0][5], q1s[e0][6], q1s[e0][7]};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    tile_colsum<false>(ew, EP, fi, fh, ss, dummy);
  }
  red0 += s; red1 += ss;
}

// end of a (non-persistent) stream kernel in training mode: every wave parks its column sums in its own tile, the first 2 * COUT
// threads add the W waves' values in wave order in fp64 and write the workgroup's partial row
template <int W, int EP, int NB>
static __device__ __forceinline__ void epi_finish_wg(const ConvP& p, float* Es, int tid, const float (&red0)[NB], const float (&red1)[NB]) {
  constexpr int COUT = NB * 32;
  static_assert(2 * COUT <= 32 * EP, "the wave's tile holds its column sums");
  const int lane = tid & 63, wv = tid >> 6, fi = lane & 31;
  float* ew = Es + wv * 32 * EP;
  if (lane < 32) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { ew[nb * 32 + fi] = red0[nb]; ew[COUT + nb * 32 + fi] = red1[nb]; }
  }
  __syncthreads();
  for (int e = tid; e < 2 * COUT; e += W * 64) {
    double s = 0.0;
    for (int w = 0; w < W; ++w) s += (double)Es[w * 32 * EP + e];
    p.red_part[(int64_t)blockIdx.x * 2 * COUT + e] = s;
  }
}

// One lane's 27 rulebook entries from the column form (9 bases + presence mask, tl_rulebook_compact)
static __device__ __forceinline__ void decode_ctab(const int32_t* __restrict__ ctab, int64_t n, int64_t row, bool rvalid, int (&idx)[27]) {
  int base[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) base[c] = rvalid ? ctab[(int64_t)c * n + row] : -1;
  const uint32_t mask = rvalid ? (uint32_t)ctab[(int64_t)9 * n + row] : 0u;
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    const uint32_t m = (mask >> (3 * c)) & 7u;
    idx[3 * c] = (m & 1u) ? base[c] : -1;
    idx[3 * c + 1] = (m & 2u) ? base[c] + (int)(m & 1u) : -1;
    idx[3 * c + 2] = (m & 4u) ? base[c] + (int)(m & 1u) + (int)((m >> 1) & 1u) : -1;
  }
}

// tl_conv_bf16.hip
int tl_launch_conv_bf16(const ConvP& p, int depth, int units, hipStream_t s);   // large levels, bf16 MFMA

// tl_conv_direct.hip
int tl_launch_conv_direct(const ConvP& p, int dtype, hipStream_t s);   // whole weight tensor resident in LDS, per-wave tiles
int tl_launch_conv_ones27(const ConvP& p, int dtype, hipStream_t s);               // every input element is 1: presence-mask table, no gather

// tl_conv_up.hip
int tl_launch_conv_up(const ConvP& p, const int32_t* child, hipStream_t s);   // 16-bit inverse conv, coarse-stationary scatter form

// tl_conv_blk.hip
int tl_launch_conv_blk(const ConvP& p, hipStream_t s);                  // 16-bit, 27 taps, 32 -> 32: rows in block-local order, staged units
#ifndef TL_F16_BUILD
int tl_launch_conv_blk_x3(const ConvP& p, hipStream_t s);               // fp32 rows, split-bf16 contraction (bf16x3), 27 taps, 32 -> 32: staged units, two 16-channel launches
int tl_conv_blk_x3_set_chunks(int n);
int tl_launch_conv_streamq_x3(const ConvP& p, int mode, hipStream_t s);          // fp32 rows, split-bf16 contraction: quad-coalesced gathers (levels 2-3)
#endif

// tl_conv_stream.hip
int tl_launch_conv_stream(const ConvP& p, int dtype, hipStream_t s);   // per-wave register gathers, weights streamed through LDS per tap

int tl_stream_set_rb(int rb);

// tl_conv_streamq.hip
int tl_launch_conv_streamq(const ConvP& p, hipStream_t s);              // bf16, Cin % 64 == 0: quad-coalesced gathers + register transposition

// tl_conv_win.hip
int tl_launch_conv_win(const ConvP& p, hipStream_t s);                  // bf16, 27 taps: dz taps of a column share one LDS-staged row window

// tl_conv_small.hip
int tl_launch_conv_small(const ConvP& p, int dtype, hipStream_t s);     // few output rows: split the tap loop over waves
int tl_launch_conv_tinycin(const ConvP& p, int dtype, hipStream_t s);   // Cin <= 8 (the 4-channel input conv)

#ifndef TL_F16_BUILD
// the float16 compilations of the same units (tl_half.h): inside them dtype TL_BF16 means "the 16-bit type"
int tl_launch_conv_direct_f16(const ConvP& p, int dtype, hipStream_t s);
int tl_launch_conv_ones27_f16(const ConvP& p, int dtype, hipStream_t s);
int tl_launch_conv_stream_f16(const ConvP& p, int dtype, hipStream_t s);
int tl_launch_conv_streamq_f16(const ConvP& p, hipStream_t s);
int tl_launch_conv_small_f16(const ConvP& p, int dtype, hipStream_t s);
int tl_launch_conv_tinycin_f16(const ConvP& p, int dtype, hipStream_t s);
int tl_launch_conv_bf16_f16(const ConvP& p, int depth, int units, hipStream_t s);
int tl_launch_conv_blk_f16(const ConvP& p, hipStream_t s);
int tl_launch_conv_up_f16(const ConvP& p, const int32_t* child, hipStream_t s);
#endif

// tl_linear_small.hip
int tl_launch_conv_tinycout(const ConvP& p, int dtype, hipStream_t s);  // K = 1, Cout <= 8 (the heads' output Linears)
