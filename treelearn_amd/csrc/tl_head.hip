// Per-point heads (fused v2p gather + output_layer BN/ReLU + both 2-layer MLPs) and row compaction.
//
// HBM-bound per-point work: one thread owns one point, holds its C-channel feature row in registers
// and walks the (wave-uniform) weights through the scalar cache, so the [N,C] gathered tensor the
// reference materialises (tree_learn.py:99) is never re-read; writing it back is optional.
#include "tl_common.h"
#include <hip/hip_bf16.h>
#include "tl_half.h"

#ifdef TL_F16_BUILD       // the float16 build of this unit (tl_half.h): tl_head_mlp_f16, reached from tl_head_mlp(dtype = TL_F16)
#define tl_head_mlp tl_head_mlp_f16
extern int g_head_mode;
#else
int g_head_mode = 0;      // developer A/B (tl_set_tuning "head_mode"): 1 = the scalar-weight kernel also for bf16 C = 32
extern "C" int tl_head_mlp_f16(const void* feats, int64_t feats_ld, int dtype, int C, const int64_t* v2p, int64_t N, const float* pro_scale,
                               const float* pro_shift, const float* w1, const float* b1, const float* w2, const float* b2, float* backbone,
                               float* logits, float* offsets, tl_stream_t stream);
#endif

namespace {

template <int C, typename T>
__global__ void __launch_bounds__(256) k_head(const T* __restrict__ feats, int64_t ld, const int64_t* __restrict__ v2p, int64_t N,
                                              const float* __restrict__ psc, const float* __restrict__ psh,
                                              const float* __restrict__ w1, const float* __restrict__ b1,
                                              const float* __restrict__ w2, const float* __restrict__ b2,
                                              float* __restrict__ backbone, float* __restrict__ logits, float* __restrict__ offsets) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = v2p[i];
    float f[C];
    if constexpr (sizeof(T) == 4) {
      const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(feats) + row * ld);
#pragma unroll
      for (int c = 0; c < C / 4; ++c) { const float4 v = src[c]; f[4 * c] = v.x; f[4 * c + 1] = v.y; f[4 * c + 2] = v.z; f[4 * c + 3] = v.w; }
    } else {
      const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const __hip_bfloat16*>(feats) + row * ld);
#pragma unroll
      for (int c = 0; c < C / 8; ++c) {
        const uint4 v = src[c];
        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { f[8 * c + 2 * q] = h16_lo(u[q]); f[8 * c + 2 * q + 1] = h16_hi(u[q]); }
      }
    }
    if (psc) {
#pragma unroll
      for (int c = 0; c < C; ++c) f[c] = fmaxf(fmaf(f[c], psc[c], psh[c]), 0.f);
    }
    if (backbone) {
      float4* dst = reinterpret_cast<float4*>(backbone + i * C);
#pragma unroll
      for (int c = 0; c < C / 4; ++c) dst[c] = make_float4(f[4 * c], f[4 * c + 1], f[4 * c + 2], f[4 * c + 3]);
    }
    float y[5] = {b2[0], b2[1], b2[2], b2[3], b2[4]};
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const float* W = w1 + hh * C * C;
      for (int j = 0; j < C; ++j) {                       // uniform index -> scalar loads of the weight row
        float h = b1[hh * C + j];
#pragma unroll
        for (int c = 0; c < C; ++c) h = fmaf(W[j * C + c], f[c], h);
        h = fmaxf(h, 0.f);
        if (hh == 0) { y[0] = fmaf(w2[0 * C + j], h, y[0]); y[1] = fmaf(w2[1 * C + j], h, y[1]); }
        else { y[2] = fmaf(w2[2 * C + j], h, y[2]); y[3] = fmaf(w2[3 * C + j], h, y[3]); y[4] = fmaf(w2[4 * C + j], h, y[4]); }
      }
    }
    logits[i * 2] = y[0]; logits[i * 2 + 1] = y[1];
    offsets[i * 3] = y[2]; offsets[i * 3 + 1] = y[3]; offsets[i * 3 + 2] = y[4];
  }
}

// bf16 features, C = 32 (the production shape): both hidden layers on the matrix cores.  The scalar-weight kernel above is
// VALU-bound (2 112 FMAs per point: 130 us of its 160 us on the config-2 tile); here a wave takes 32 points at a time and computes
// H^T = W1 X^T with 32x32x16 bf16 MFMAs -- A = the hidden layer's weights (hidden unit x channel, constant fragments in
// registers), B = the points' BatchNorm+ReLU'd feature rows (point x channel: lane (n, h) gathers 16-byte pieces h and 2 + h of
// row v2p[n]) -- so that afterwards lane (n, h) holds 16 hidden units of point n; the output layer is 16 FMAs per output on
// those, and one cross-half add.  Activations and W1 are rounded to bf16 for the MFMA (the backbone output stays fp32).
typedef float hf32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t hu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t head_pack_bf16x2(float lo, float hi) { return h16_pack2(lo, hi); }      // round-to-nearest-even, this build's 16-bit type

__global__ void __launch_bounds__(256) k_head_mfma32(const __hip_bfloat16* __restrict__ feats, int64_t ld, const int64_t* __restrict__ v2p, int64_t N,
                                                     const float* __restrict__ psc, const float* __restrict__ psh,
                                                     const float* __restrict__ w1, const float* __restrict__ b1,
                                                     const float* __restrict__ w2, const float* __restrict__ b2,
                                                     float* __restrict__ backbone, float* __restrict__ logits, float* __restrict__ offsets) {
  constexpr int C = 32;
  __shared__ float4 Tb[2][2][16];                            // [h][head][r] = {b1[j], w2[k0][j], w2[k1][j], w2[k2][j]}, j = (r&3) + 8 (r>>2) + 4 h
  __shared__ float4 Fs[4][32][9];                            // per wave: the tile's fp32 rows, [point][8 float4 + 1 pad], for whole-row stores
  const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
  for (int e = threadIdx.x; e < 64; e += 256) {
    const int hh_ = e >> 5, r = e & 15, h_ = (e >> 4) & 1;
    const int j = (r & 3) + 8 * (r >> 2) + 4 * h_;
    Tb[h_][hh_][r] = hh_ == 0 ? make_float4(b1[j], w2[0 * C + j], w2[1 * C + j], 0.f)
                              : make_float4(b1[C + j], w2[2 * C + j], w2[3 * C + j], w2[4 * C + j]);
  }
  // A fragments: lane (m = hidden unit, h) of k-step s holds W1[head][m][16 s + 8 h .. + 8]
  hu32x4 afrag[2][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float* wr = w1 + ((int64_t)hh * C + n) * C + 16 * s + 8 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q) afrag[hh][s][q] = head_pack_bf16x2(wr[2 * q], wr[2 * q + 1]);
    }
  // the lane's 16 channels (16 s + 8 h + q) of the output_layer affine
  float sc[2][8], sh[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int q = 0; q < 8; ++q) { sc[s][q] = psc ? psc[16 * s + 8 * h + q] : 1.f; sh[s][q] = psc ? psh[16 * s + 8 * h + q] : 0.f; }
  const float bo0 = b2[0], bo1 = b2[1], bo2 = b2[2], bo3 = b2[3], bo4 = b2[4];
  __syncthreads();

  const int64_t ntiles = (N + 31) / 32;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t p = tile * 32 + n;
    const bool pv = p < N;
    const int64_t row = pv ? v2p[p] : 0;
    const uint4* src = reinterpret_cast<const uint4*>(feats + row * ld + 8 * h);
    hu32x4 bfrag[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint4 v = src[2 * s];                            // channels 16 s + 8 h .. + 8
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
      float f[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { f[2 * q] = h16_lo(u[q]); f[2 * q + 1] = h16_hi(u[q]); }
      if (psc) {
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = fmaxf(fmaf(f[q], sc[s][q], sh[s][q]), 0.f);
      }
      if (backbone) {
        float4* fs = &Fs[threadIdx.x >> 6][n][4 * s + 2 * h];
        fs[0] = make_float4(f[0], f[1], f[2], f[3]); fs[1] = make_float4(f[4], f[5], f[6], f[7]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) bfrag[s][q] = head_pack_bf16x2(f[2 * q], f[2 * q + 1]);
    }
    if (backbone) {                                          // eight lanes write one point's 128 B: every store instruction covers 8 whole rows
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = 8 * i + (lane >> 3), sl = lane & 7;
        const int64_t q = tile * 32 + rr;
        if (q < N) reinterpret_cast<float4*>(backbone + q * C)[sl] = Fs[threadIdx.x >> 6][rr][sl];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    float y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f, y4 = 0.f;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      hf32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
        acc = h16_mfma(afrag[hh][s], bfrag[s], acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 t = Tb[h][hh][r];
        const float hv = fmaxf(acc[r] + t.x, 0.f);
        if (hh == 0) { y0 = fmaf(t.y, hv, y0); y1 = fmaf(t.z, hv, y1); }
        else { y2 = fmaf(t.y, hv, y2); y3 = fmaf(t.z, hv, y3); y4 = fmaf(t.w, hv, y4); }
      }
    }
    y0 += __shfl_xor(y0, 32); y1 += __shfl_xor(y1, 32); y2 += __shfl_xor(y2, 32); y3 += __shfl_xor(y3, 32); y4 += __shfl_xor(y4, 32);
    if (pv) {
      if (h == 0) { logits[p * 2] = y0 + bo0; logits[p * 2 + 1] = y1 + bo1; }
      else { offsets[p * 3] = y2 + bo2; offsets[p * 3 + 1] = y3 + bo3; offsets[p * 3 + 2] = y4 + bo4; }
    }
  }
}

template <int C>
int launch_head(const void* feats, int64_t ld, int dtype, const int64_t* v2p, int64_t N, const float* psc, const float* psh,
                const float* w1, const float* b1, const float* w2, const float* b2, float* bb, float* lg, float* of, hipStream_t s) {
  const unsigned g = tl_grid(N, 256);
  if constexpr (C == 32) {
    if (dtype == TL_BF16 && g_head_mode != 1) {
      const int64_t need = tl_cdiv(tl_cdiv(N, 32), 4);
      k_head_mfma32<<<(unsigned)(need < 2048 ? need : 2048), 256, 0, s>>>((const __hip_bfloat16*)feats, ld, v2p, N, psc, psh, w1, b1, w2, b2, bb, lg, of);
      return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
    }
  }
  if (dtype == TL_F32) k_head<C, float><<<g, 256, 0, s>>>((const float*)feats, ld, v2p, N, psc, psh, w1, b1, w2, b2, bb, lg, of);
  else k_head<C, __hip_bfloat16><<<g, 256, 0, s>>>((const __hip_bfloat16*)feats, ld, v2p, N, psc, psh, w1, b1, w2, b2, bb, lg, of);
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

#ifndef TL_F16_BUILD
// ---------------------------------------------------------------- stable row compaction
constexpr int kItems = 8, kTile = 256 * kItems;

__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wsum[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t inc = v;
  for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, off); if (lane >= off) inc += t; }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < 4; ++w) { if (w < wid) base += wsum[w]; tot += wsum[w]; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(256) k_mask_partials(const uint8_t* __restrict__ m, int64_t n, int32_t* __restrict__ part) {
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  uint32_t s = 0;
  for (int j = 0; j < kItems; ++j) if (base + j < n) s += m[base + j] != 0;
  uint32_t tot; block_scan(s, &tot);
  if (threadIdx.x == 0) part[blockIdx.x] = (int32_t)tot;
}
__global__ void __launch_bounds__(256) k_mask_scan(int32_t* __restrict__ part, int64_t nb, int32_t* __restrict__ count) {
  uint32_t carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += 256) {
    const int64_t i = b0 + threadIdx.x;
    const uint32_t v = i < nb ? (uint32_t)part[i] : 0u;
    uint32_t tot; const uint32_t ex = block_scan(v, &tot);
    if (i < nb) part[i] = (int32_t)(carry + ex);
    carry += tot;
  }
  if (threadIdx.x == 0) *count = (int32_t)carry;
}
__global__ void __launch_bounds__(256) k_mask_scatter(const float* __restrict__ in, int C, const uint8_t* __restrict__ m, int64_t n,
                                                      const int32_t* __restrict__ part, float* __restrict__ out) {
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  uint32_t s = 0; bool keep[kItems];
  for (int j = 0; j < kItems; ++j) { keep[j] = (base + j < n) && m[base + j] != 0; s += keep[j]; }
  uint32_t tot; uint32_t pos = block_scan(s, &tot) + (uint32_t)part[blockIdx.x];
  for (int j = 0; j < kItems; ++j) if (keep[j]) { for (int c = 0; c < C; ++c) out[(int64_t)pos * C + c] = in[(base + j) * C + c]; ++pos; }
}

#endif  // !TL_F16_BUILD

}  // namespace

extern "C" {

int tl_head_mlp(const void* feats, int64_t feats_ld, int dtype, int C, const int64_t* v2p, int64_t N, const float* pro_scale,
                const float* pro_shift, const float* w1, const float* b1, const float* w2, const float* b2, float* backbone,
                float* logits, float* offsets, tl_stream_t stream) {
  if (!feats || !v2p || !w1 || !b1 || !w2 || !b2 || !logits || !offsets || N <= 0) return TL_ERR_ARG;
  if ((pro_scale == nullptr) != (pro_shift == nullptr)) return TL_ERR_ARG;
#ifndef TL_F16_BUILD
  if (dtype == TL_F16)      // the float16 compilation of this unit; inside it "TL_BF16" means "the 16-bit type"
    return tl_head_mlp_f16(feats, feats_ld, TL_BF16, C, v2p, N, pro_scale, pro_shift, w1, b1, w2, b2, backbone, logits, offsets, stream);
#endif
  if (dtype != TL_F32 && dtype != TL_BF16) return TL_ERR_ARG;
  if (feats_ld % 8 != 0 || ((uintptr_t)feats) % 16 != 0) return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  switch (C) {
    case 8: return launch_head<8>(feats, feats_ld, dtype, v2p, N, pro_scale, pro_shift, w1, b1, w2, b2, backbone, logits, offsets, s);
    case 16: return launch_head<16>(feats, feats_ld, dtype, v2p, N, pro_scale, pro_shift, w1, b1, w2, b2, backbone, logits, offsets, s);
    case 32: return launch_head<32>(feats, feats_ld, dtype, v2p, N, pro_scale, pro_shift, w1, b1, w2, b2, backbone, logits, offsets, s);
    case 64: return launch_head<64>(feats, feats_ld, dtype, v2p, N, pro_scale, pro_shift, w1, b1, w2, b2, backbone, logits, offsets, s);
  }
  return TL_ERR_UNSUPPORTED;
}

#ifndef TL_F16_BUILD
int64_t tl_compact_ws_words(int64_t n) { return tl_cdiv(n, kTile) + 1; }

int tl_compact_rows(const float* in, int C, const uint8_t* mask, int64_t n, float* out, int32_t* count, int32_t* ws, tl_stream_t stream) {
  if (!in || !mask || !out || !count || !ws || C <= 0 || n <= 0) return TL_ERR_ARG;
  const int64_t nb = tl_cdiv(n, kTile);
  hipStream_t s = tl_s(stream);
  k_mask_partials<<<(unsigned)nb, 256, 0, s>>>(mask, n, ws);
  k_mask_scan<<<1, 256, 0, s>>>(ws, nb, count);
  k_mask_scatter<<<(unsigned)nb, 256, 0, s>>>(in, C, mask, n, ws, out);
  TL_CHECK_LAUNCH();
  return TL_OK;
}
#endif  // !TL_F16_BUILD

}  // extern "C"
