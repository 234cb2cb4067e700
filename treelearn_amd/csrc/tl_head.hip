// Per-point heads (fused v2p gather + output_layer BN/ReLU + both 2-layer MLPs) and row compaction.
//
// HBM-bound per-point work: one thread owns one point, holds its C-channel feature row in registers
// and walks the (wave-uniform) weights through the scalar cache, so the [N,C] gathered tensor the
// reference materialises (tree_learn.py:99) is never re-read; writing it back is optional.
#include "tl_common.h"
#include <hip/hip_bf16.h>

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
        for (int q = 0; q < 4; ++q) { f[8 * c + 2 * q] = __uint_as_float(u[q] << 16); f[8 * c + 2 * q + 1] = __uint_as_float(u[q] & 0xFFFF0000u); }
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

template <int C>
int launch_head(const void* feats, int64_t ld, int dtype, const int64_t* v2p, int64_t N, const float* psc, const float* psh,
                const float* w1, const float* b1, const float* w2, const float* b2, float* bb, float* lg, float* of, hipStream_t s) {
  const unsigned g = tl_grid(N, 256);
  if (dtype == TL_F32) k_head<C, float><<<g, 256, 0, s>>>((const float*)feats, ld, v2p, N, psc, psh, w1, b1, w2, b2, bb, lg, of);
  else k_head<C, __hip_bfloat16><<<g, 256, 0, s>>>((const __hip_bfloat16*)feats, ld, v2p, N, psc, psh, w1, b1, w2, b2, bb, lg, of);
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

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

}  // namespace

extern "C" {

int tl_head_mlp(const void* feats, int64_t feats_ld, int dtype, int C, const int64_t* v2p, int64_t N, const float* pro_scale,
                const float* pro_shift, const float* w1, const float* b1, const float* w2, const float* b2, float* backbone,
                float* logits, float* offsets, tl_stream_t stream) {
  if (!feats || !v2p || !w1 || !b1 || !w2 || !b2 || !logits || !offsets || N <= 0) return TL_ERR_ARG;
  if ((pro_scale == nullptr) != (pro_shift == nullptr)) return TL_ERR_ARG;
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

}  // extern "C"
