// Voxel -> point feature gather and its gradient (reference tree_learn/model/tree_learn.py:98 `backbone_feats = output.features[v2p_map]`,
// differentiated by tools/training/train.py:40).  ATen's index kernel took 0.8 ms for the 3.7 M x 32 bf16 rows of a config-3 batch and
// its deterministic index_put backward (sort + segmented walk inside) 1.4 ms; here both are plain row copies over 16-B pieces:
//   tl_gather_rows     : out[p] = in[idx[p]]                                   (idx < 0 counts from the end, as torch indexing does)
//   tl_scatter_add_rows: gin[v] = sum of g[p] over the points p of voxel v, added in ascending p (the caller passes the stable argsort
//                        of idx once per batch), fp32 accumulation, every voxel row written exactly once -- deterministic, no atomics.
#include "tl_conv_internal.h"
#include "tl_f16_train.h"

namespace {

template <bool BF16>
__global__ void __launch_bounds__(256) k_gather_rows(const char* __restrict__ in, int64_t in_ld_b, int PR, int64_t n_rows, const int64_t* __restrict__ idx, int64_t N,
                                                     char* __restrict__ out, int64_t out_ld_b) {
  const int64_t total = N * PR;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t p = t / PR; const int pc = (int)(t % PR);
    int64_t r = idx[p];
    if (r < 0) r += n_rows;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r >= 0 && r < n_rows) v = *reinterpret_cast<const u32x4*>(in + r * in_ld_b + pc * 16);
    *reinterpret_cast<u32x4*>(out + p * out_ld_b + pc * 16) = v;
  }
}

template <bool BF16>
__global__ void __launch_bounds__(256) k_scatter_add_rows(const char* __restrict__ g, int64_t g_ld_b, int PR, const int64_t* __restrict__ order,
                                                          const int64_t* __restrict__ sidx, int64_t N, int64_t n_rows, char* __restrict__ gin, int64_t gin_ld_b) {
  const int64_t total = N * PR;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t j = t / PR; const int pc = (int)(t % PR);
    const int64_t v0 = sidx[j];
    if (j > 0 && sidx[j - 1] == v0) continue;                    // not the first point of its voxel
    int64_t v = v0 < 0 ? v0 + n_rows : v0;
    if (v < 0 || v >= n_rows) continue;
    const u32x4 q = *reinterpret_cast<const u32x4*>(g + order[j] * g_ld_b + pc * 16);
    int64_t jj = j + 1;
    if (jj >= N || sidx[jj] != v0) {                              // one point in the voxel (the common case): a plain copy
      *reinterpret_cast<u32x4*>(gin + v * gin_ld_b + pc * 16) = q;
      continue;
    }
    if constexpr (BF16) {
      float a[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[2 * i] = bf16_lo(q[i]); a[2 * i + 1] = bf16_hi(q[i]); }
      for (; jj < N && sidx[jj] == v0; ++jj) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(g + order[jj] * g_ld_b + pc * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[2 * i] += bf16_lo(w[i]); a[2 * i + 1] += bf16_hi(w[i]); }
      }
      u32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(a[2 * i], a[2 * i + 1]);
      *reinterpret_cast<u32x4*>(gin + v * gin_ld_b + pc * 16) = o;
    } else {
      f32x4 a = __builtin_bit_cast(f32x4, q);
      for (; jj < N && sidx[jj] == v0; ++jj) a += *reinterpret_cast<const f32x4*>(g + order[jj] * g_ld_b + pc * 16);
      *reinterpret_cast<f32x4*>(gin + v * gin_ld_b + pc * 16) = a;
    }
  }
}

}  // namespace

extern "C" {

int tl_gather_rows(const void* in, int64_t in_ld, int dtype, int C, int64_t n_rows, const int64_t* idx, int64_t N, void* out, int64_t out_ld, tl_stream_t stream) {
#ifndef TL_F16_BUILD
  if (dtype == TL_F16) return tl_gather_rows_f16(in, in_ld, TL_BF16, C, n_rows, idx, N, out, out_ld, stream);
#endif
  if (!in || !idx || !out || C <= 0 || n_rows <= 0 || N <= 0 || (dtype != TL_F32 && dtype != TL_BF16)) return TL_ERR_ARG;
  const int eb = dtype == TL_BF16 ? 2 : 4, epv = 16 / eb;
  if (C % epv || in_ld % epv || out_ld % epv || ((uintptr_t)in) % 16 || ((uintptr_t)out) % 16) return TL_ERR_UNSUPPORTED;
  const int PR = C / epv;
  const unsigned g = tl_grid(N * PR, 256);
  if (dtype == TL_BF16) k_gather_rows<true><<<g, 256, 0, tl_s(stream)>>>((const char*)in, in_ld * eb, PR, n_rows, idx, N, (char*)out, out_ld * eb);
  else k_gather_rows<false><<<g, 256, 0, tl_s(stream)>>>((const char*)in, in_ld * eb, PR, n_rows, idx, N, (char*)out, out_ld * eb);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_scatter_add_rows(const void* g, int64_t g_ld, int dtype, int C, const int64_t* order, const int64_t* sorted_idx, int64_t N, int64_t n_rows, void* gin,
                        int64_t gin_ld, tl_stream_t stream) {
#ifndef TL_F16_BUILD
  if (dtype == TL_F16) return tl_scatter_add_rows_f16(g, g_ld, TL_BF16, C, order, sorted_idx, N, n_rows, gin, gin_ld, stream);
#endif
  if (!g || !order || !sorted_idx || !gin || C <= 0 || n_rows <= 0 || N <= 0 || (dtype != TL_F32 && dtype != TL_BF16)) return TL_ERR_ARG;
  const int eb = dtype == TL_BF16 ? 2 : 4, epv = 16 / eb;
  if (C % epv || g_ld % epv || gin_ld != C || ((uintptr_t)g) % 16 || ((uintptr_t)gin) % 16) return TL_ERR_UNSUPPORTED;
  hipStream_t s = tl_s(stream);
  if (hipMemsetAsync(gin, 0, (size_t)n_rows * C * eb, s) != hipSuccess) return TL_ERR_LAUNCH;       // voxels without a point
  const int PR = C / epv;
  const unsigned gr = tl_grid(N * PR, 256);
  if (dtype == TL_BF16) k_scatter_add_rows<true><<<gr, 256, 0, s>>>((const char*)g, g_ld * eb, PR, order, sorted_idx, N, n_rows, (char*)gin, gin_ld * eb);
  else k_scatter_add_rows<false><<<gr, 256, 0, s>>>((const char*)g, g_ld * eb, PR, order, sorted_idx, N, n_rows, (char*)gin, gin_ld * eb);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

}  // extern "C"
