// BatchNorm1d in TRAINING mode (+ ReLU) over the rows of a feature matrix, forward and backward -- the `norm_fn(C), nn.ReLU()` pair
// that sits in front of every one of the U-Net's 65 convs and in both heads (reference tree_learn/model/blocks.py:55-70,102-123,
// tree_learn.py:34-46; BatchNorm1d(eps=1e-4, momentum=0.1) normalises over all active voxels of the batch).
//
// HBM-bound elementwise / reduction work.  Everything is deterministic: a block owns a fixed contiguous range of rows, every
// thread accumulates its 4 channels over its rows in fp64, the row lanes of a block are added in lane order through LDS, and the
// finishing kernel adds the block partials in a fixed order (no atomics).
//
//   forward :  tl_bn_train_stats   x -> per-channel mean / biased variance (one read of x, fp64 sum and sum of squares),
//                                  scale = gamma * rstd, shift = beta - mean * scale, running statistics updated in place
//              tl_affine_relu      y = relu(x * scale + shift)            (tl_conv.hip; the eval path's kernel)
//   backward:  tl_bn_train_bwd_reduce  g = dy * [y > 0];  sum_g, sum_g_xhat per channel  (= dbeta, dgamma)
//              tl_bn_train_bwd_apply   dx = scale * (g - sum_g / n - xhat * sum_g_xhat / n)
#include "tl_conv_internal.h"
#include "tl_f16_train.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;

template <typename T>
static __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
  const f32x4 q = *reinterpret_cast<const f32x4*>(p);
  v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
}
template <>
__device__ __forceinline__ void load4<__hip_bfloat16>(const __hip_bfloat16* p, float (&v)[4]) {
  const uint2 q = *reinterpret_cast<const uint2*>(p);
  v[0] = bf16_lo(q.x); v[1] = bf16_hi(q.x); v[2] = bf16_lo(q.y); v[3] = bf16_hi(q.y);
}
static __device__ __forceinline__ void store4(float* p, const float (&v)[4]) { *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]}; }
static __device__ __forceinline__ void store4(__hip_bfloat16* p, const float (&v)[4]) {
  *reinterpret_cast<uint2*>(p) = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
}

// block b owns rows [b * rpb, min(n, (b + 1) * rpb)); thread t: channel group t % cg (4 channels), row lane t / cg of RL
struct Part {
  int cg, rl;
  int64_t rpb;
};
static inline Part partition(int64_t n, int C, int& blocks) {
  Part p;
  p.cg = C / 4; p.rl = kThreads / p.cg;
  int64_t b = tl_cdiv(n, (int64_t)p.rl * 8);
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  p.rpb = tl_cdiv(n, b);
  blocks = (int)tl_cdiv(n, p.rpb);
  return p;
}

// NQ fp64 quantities per channel, summed over the block's rows; lanes added in lane order -> part[b][q][C]
template <int NQ>
static __device__ __forceinline__ void block_reduce_store(double (&acc)[NQ][4], const Part& pt, int C, double* __restrict__ part) {
  extern __shared__ double sred[];                       // [rl][NQ][C]
  const int t = threadIdx.x, cgi = t % pt.cg, lane = t / pt.cg;
  if (lane < pt.rl) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) sred[((int64_t)lane * NQ + q) * C + cgi * 4 + c] = acc[q][c];
  }
  __syncthreads();
  for (int e = t; e < NQ * C; e += kThreads) {
    double s = 0.0;
    for (int l = 0; l < pt.rl; ++l) s += sred[(int64_t)l * NQ * C + e];
    part[(int64_t)blockIdx.x * NQ * C + e] = s;
  }
}

template <typename T>
__global__ void __launch_bounds__(kThreads) k_bn_stats(const T* __restrict__ x, int64_t ld, int64_t n, int C, Part pt, double* __restrict__ part) {
  const int t = threadIdx.x, cgi = t % pt.cg, lane = t / pt.cg;
  double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  if (lane < pt.rl) {
    const int64_t r0 = (int64_t)blockIdx.x * pt.rpb, r1 = min(n, r0 + pt.rpb);
#pragma unroll 4
    for (int64_t r = r0 + lane; r < r1; r += pt.rl) {
      float v[4];
      load4<T>(x + r * ld + cgi * 4, v);
#pragma unroll
      for (int c = 0; c < 4; ++c) { const double d = (double)v[c]; acc[0][c] += d; acc[1][c] = fma(d, d, acc[1][c]); }
    }
  }
  block_reduce_store<2>(acc, pt, C, part);
}

// fixed-order sum of the block partials of one (quantity, channel) by a 256-thread workgroup: thread t adds partials t, t + 256, ...
// (four independent sums, so many loads are in flight: a conv epilogue leaves up to 15 000 partial rows), the 256 thread sums go
// through a fixed LDS tree -- deterministic.  Every thread returns the total.
constexpr int kFin = 256;
static __device__ __forceinline__ double sum_partials(const double* __restrict__ part, int blocks, int64_t stride, int64_t off, double* red) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int b = threadIdx.x;
  for (; b + 3 * kFin < blocks; b += 4 * kFin) {
    s0 += part[(int64_t)b * stride + off]; s1 += part[(int64_t)(b + kFin) * stride + off];
    s2 += part[(int64_t)(b + 2 * kFin) * stride + off]; s3 += part[(int64_t)(b + 3 * kFin) * stride + off];
  }
  for (; b < blocks; b += kFin) s0 += part[(int64_t)b * stride + off];
  __syncthreads();                                         // red is reused by consecutive calls
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int w = kFin / 2; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  return red[0];
}

// one wave per channel: block partials -> mean, biased var, scale / shift, running statistics
__global__ void __launch_bounds__(kFin) k_bn_stats_finish(const double* __restrict__ part, int blocks, int64_t n, int C, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, float momentum, float* __restrict__ mean,
                                                        float* __restrict__ rstd, float* __restrict__ scale, float* __restrict__ shift,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        int64_t* __restrict__ num_batches_tracked) {
  __builtin_amdgcn_s_setprio(3);                              // a short kernel on the critical path, often next to a persistent weight-gradient kernel of the side stream
  __shared__ double red[kFin];
  const int c = blockIdx.x;
  const double s = sum_partials(part, blocks, 2 * (int64_t)C, c, red), ss = sum_partials(part, blocks, 2 * (int64_t)C, C + c, red);
  if (threadIdx.x != 0) return;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  const double m = s / (double)n;
  double var = ss / (double)n - m * m;
  if (var < 0.0) var = 0.0;
  const float mf = (float)m, vf = (float)var;
  const float rs = 1.0f / sqrtf(vf + eps);
  mean[c] = mf; rstd[c] = rs;
  const float sc = gamma[c] * rs;
  scale[c] = sc; shift[c] = beta[c] - mf * sc;
  if (running_mean) {
    const float unbiased = n > 1 ? (float)(var * (double)n / (double)(n - 1)) : vf;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mf;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// g = dy * [relu ? x * scale + shift > 0 : 1];  sums of g and g * xhat
template <typename TX, typename TG>
__global__ void __launch_bounds__(kThreads) k_bn_bwd_reduce(const TX* __restrict__ x, int64_t ld, const TG* __restrict__ dy, int64_t dld, int64_t n, int C,
                                                            Part pt, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                            double* __restrict__ part) {
  const int t = threadIdx.x, cgi = t % pt.cg, lane = t / pt.cg;
  double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  if (lane < pt.rl) {
    float mu[4], rs[4], sc[4], sh[4];
    load4<float>(mean + cgi * 4, mu); load4<float>(rstd + cgi * 4, rs); load4<float>(scale + cgi * 4, sc); load4<float>(shift + cgi * 4, sh);
    const int64_t r0 = (int64_t)blockIdx.x * pt.rpb, r1 = min(n, r0 + pt.rpb);
#pragma unroll 4
    for (int64_t r = r0 + lane; r < r1; r += pt.rl) {
      float v[4], g[4];
      load4<TX>(x + r * ld + cgi * 4, v);
      load4<TG>(dy + r * dld + cgi * 4, g);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float gg = (relu && !(fmaf(v[c], sc[c], sh[c]) > 0.f)) ? 0.f : g[c];
        const float xh = (v[c] - mu[c]) * rs[c];
        acc[0][c] += (double)gg; acc[1][c] = fma((double)gg, (double)xh, acc[1][c]);
      }
    }
  }
  block_reduce_store<2>(acc, pt, C, part);
}

__global__ void __launch_bounds__(kFin) k_bn_bwd_finish(const double* __restrict__ part, int blocks, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __builtin_amdgcn_s_setprio(3);                              // a short kernel on the critical path, often next to a persistent weight-gradient kernel of the side stream
  __shared__ double red[kFin];
  const int c = blockIdx.x;
  const double s = sum_partials(part, blocks, 2 * (int64_t)C, c, red), sx = sum_partials(part, blocks, 2 * (int64_t)C, C + c, red);
  if (threadIdx.x == 0) { dbeta[c] = (float)s; dgamma[c] = (float)sx; }
}

template <typename TX, typename TG>
__global__ void __launch_bounds__(kThreads) k_bn_bwd_apply(const TX* __restrict__ x, int64_t ld, const TG* __restrict__ dy, int64_t dld, int64_t n, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu, const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, TX* __restrict__ dx, int64_t xld,
                                                           const TX* __restrict__ add, int64_t ald) {
  const int cg = C / 4;
  const int64_t total = n * cg;
  const float inv_n = 1.0f / (float)n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / cg; const int c0 = (int)(e % cg) * 4;
    float v[4], g[4], mu[4], rs[4], sc[4], sh[4], dg[4], db[4], o[4];
    load4<TX>(x + r * ld + c0, v); load4<TG>(dy + r * dld + c0, g);
    load4<float>(mean + c0, mu); load4<float>(rstd + c0, rs); load4<float>(scale + c0, sc); load4<float>(shift + c0, sh);
    load4<float>(dgamma + c0, dg); load4<float>(dbeta + c0, db);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float gg = (relu && !(fmaf(v[c], sc[c], sh[c]) > 0.f)) ? 0.f : g[c];
      const float xh = (v[c] - mu[c]) * rs[c];
      o[c] = sc[c] * (gg - db[c] * inv_n - xh * dg[c] * inv_n);
    }
    if (add) {                                             // the gradient that reaches x along its other use (residual / skip), accumulated here
      float a[4];
      load4<TX>(add + r * ald + c0, a);
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] += a[c];
    }
    store4(dx + r * xld + c0, o);
  }
}

// dx over 8-channel vectors with the thread's channel group fixed (the grid stride is a multiple of the vectors per row): the six
// per-channel constants live in registers as two fused coefficients, one 16-B load per operand and step.
//   dx = a * g + b - c * x        a = scale,  c = scale * rstd * dgamma / n,  b = -scale * dbeta / n + c * mean
// MASK: g = dy * [x * scale + shift > 0] is formed here (dy raw); otherwise dy is already masked (conv epilogue TL_EPI_BN_BWD)
template <bool XB, bool GB, bool MASK>
__global__ void __launch_bounds__(kThreads) k_bn_bwd_apply_v8(const void* __restrict__ x, int64_t ld, const void* __restrict__ dy, int64_t dld, int64_t n, int C,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                              void* __restrict__ dx, int64_t xld, const void* __restrict__ add, int64_t ald) {
  const int vpr = C >> 3;
  const int64_t total = n * vpr, stride = (int64_t)gridDim.x * kThreads;
  int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (v >= total) return;
  const int c0 = (int)(v % vpr) * 8;
  const float inv_n = 1.0f / (float)n;
  float ca[8], cb[8], cc[8], sc[8], sh[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    sc[q] = scale[c0 + q]; sh[q] = shift[c0 + q];
    ca[q] = sc[q];
    cc[q] = sc[q] * rstd[c0 + q] * dgamma[c0 + q] * inv_n;
    cb[q] = cc[q] * mean[c0 + q] - sc[q] * dbeta[c0 + q] * inv_n;
  }
  auto ld8 = [](const void* base, int64_t elem, bool bf, float (&o)[8]) __attribute__((always_inline)) {
    if (bf) {
      const u32x4 q4 = *reinterpret_cast<const u32x4*>((const uint16_t*)base + elem);
#pragma unroll
      for (int q = 0; q < 4; ++q) { o[2 * q] = bf16_lo(q4[q]); o[2 * q + 1] = bf16_hi(q4[q]); }
    } else {
      const f32x4 a = *reinterpret_cast<const f32x4*>((const float*)base + elem), b = *reinterpret_cast<const f32x4*>((const float*)base + elem + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { o[q] = a[q]; o[q + 4] = b[q]; }
    }
  };
  for (; v < total; v += stride) {
    const int64_t r = v / vpr;
    float xv[8], g[8], o[8];
    ld8(x, r * ld + c0, XB, xv); ld8(dy, r * dld + c0, GB, g);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float gg = g[q];
      if constexpr (MASK) { if (!(fmaf(xv[q], sc[q], sh[q]) > 0.f)) gg = 0.f; }
      o[q] = fmaf(ca[q], gg, fmaf(-cc[q], xv[q], cb[q]));
    }
    if (add) {
      float a[8];
      ld8(add, r * ald + c0, XB, a);
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] += a[q];
    }
    if constexpr (XB) {
      u32x4 w;
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = pack_bf16x2(o[2 * q], o[2 * q + 1]);
      *reinterpret_cast<u32x4*>((uint16_t*)dx + r * xld + c0) = w;
    } else {
      f32x4* d = reinterpret_cast<f32x4*>((float*)dx + r * xld + c0);
      d[0] = f32x4{o[0], o[1], o[2], o[3]}; d[1] = f32x4{o[4], o[5], o[6], o[7]};
    }
  }
}

// launch the vector form if the views allow it; false = caller uses the 4-channel kernel
static bool launch_apply_v8(const void* x, int64_t ld, int x_dtype, const void* dy, int64_t dld, int dy_dtype, int64_t n, int C, const float* mean, const float* rstd,
                            const float* scale, const float* shift, bool mask, const float* dgamma, const float* dbeta, void* dx, int64_t xld, const void* add,
                            int64_t ald, hipStream_t s) {
  if (C % 8 || ld % 8 || dld % 8 || xld % 8 || (add && ald % 8) || ((uintptr_t)x) % 16 || ((uintptr_t)dy) % 16 || ((uintptr_t)dx) % 16 || (add && ((uintptr_t)add) % 16))
    return false;
  const int vpr = C / 8;
  int64_t gv = tl_cdiv(n * vpr, kThreads * 4);
  if (gv > 256 * 16) gv = 256 * 16;
  gv = tl_cdiv(gv, (int64_t)vpr) * vpr;                        // grid * 256 is a multiple of the vectors per row
  const bool xb = x_dtype == TL_BF16, gb = dy_dtype == TL_BF16;
#define TL_AP(XB_, GB_, M_) k_bn_bwd_apply_v8<XB_, GB_, M_><<<(unsigned)gv, kThreads, 0, s>>>(x, ld, dy, dld, n, C, mean, rstd, scale, shift, dgamma, dbeta, dx, xld, add, ald)
  if (xb && gb) { if (mask) TL_AP(true, true, true); else TL_AP(true, true, false); }
  else if (xb && !gb) { if (mask) TL_AP(true, false, true); else TL_AP(true, false, false); }
  else if (!xb && gb) { if (mask) TL_AP(false, true, true); else TL_AP(false, true, false); }
  else { if (mask) TL_AP(false, false, true); else TL_AP(false, false, false); }
#undef TL_AP
  return true;
}

}  // namespace

extern "C" {

int tl_bn_train_finish(const double* part, int64_t nparts, int64_t n, int C, const float* gamma, const float* beta, float eps, float momentum, float* mean,
                       float* rstd, float* scale, float* shift, float* running_mean, float* running_var, int64_t* num_batches_tracked, tl_stream_t stream) {
  if (!part || nparts <= 0 || nparts > 0x7FFFFFFF || !gamma || !beta || !mean || !rstd || !scale || !shift || n <= 0 || C <= 0) return TL_ERR_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return TL_ERR_ARG;
  k_bn_stats_finish<<<C, kFin, 0, tl_s(stream)>>>(part, (int)nparts, n, C, gamma, beta, eps, momentum, mean, rstd, scale, shift, running_mean, running_var,
                                              num_batches_tracked);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_bn_train_bwd_from_parts(const void* x, int64_t ld, int x_dtype, const void* g, int64_t gld, int g_dtype, int64_t n, int C, const float* mean,
                               const float* rstd, const float* scale, const float* shift, const double* part, int64_t nparts, float* dgamma, float* dbeta,
                               void* dx, int64_t xld, const void* dx_add, int64_t ald, tl_stream_t stream) {
  if (!x || !g || !mean || !rstd || !scale || !shift || !part || nparts <= 0 || nparts > 0x7FFFFFFF || !dgamma || !dbeta || !dx || n <= 0 || C <= 0) return TL_ERR_ARG;
#ifndef TL_F16_BUILD
  if (x_dtype == TL_F16 || g_dtype == TL_F16) {
    if (x_dtype == TL_BF16 || g_dtype == TL_BF16) return TL_ERR_ARG;            // one 16-bit type per call
    return tl_bn_train_bwd_from_parts_f16(x, ld, tl_f16_code(x_dtype), g, gld, tl_f16_code(g_dtype), n, C, mean, rstd, scale, shift, part, nparts, dgamma, dbeta, dx, xld,
                                          dx_add, ald, stream);
  }
#endif
  if ((x_dtype != TL_F32 && x_dtype != TL_BF16) || (g_dtype != TL_F32 && g_dtype != TL_BF16)) return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  k_bn_bwd_finish<<<C, kFin, 0, s>>>(part, (int)nparts, C, dgamma, dbeta);
  TL_CHECK_LAUNCH();
  if (!launch_apply_v8(x, ld, x_dtype, g, gld, g_dtype, n, C, mean, rstd, scale, shift, false, dgamma, dbeta, dx, xld, dx_add, ald, s)) return TL_ERR_UNSUPPORTED;
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int64_t tl_bn_ws_doubles(int64_t n, int C) {
  if (n <= 0 || C <= 0 || C % 4 || C > 4 * kThreads) return 0;
  int blocks;
  partition(n, C, blocks);
  return (int64_t)blocks * 2 * C;
}

int tl_bn_train_stats(const void* x, int64_t ld, int64_t n, int C, int dtype, const float* gamma, const float* beta, float eps, float momentum,
                      double* ws, float* mean, float* rstd, float* scale, float* shift, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, tl_stream_t stream) {
  if (!x || !gamma || !beta || !ws || !mean || !rstd || !scale || !shift || n <= 0 || C <= 0 || C % 4 || C > 4 * kThreads || ld % 4) return TL_ERR_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return TL_ERR_ARG;
#ifndef TL_F16_BUILD
  if (dtype == TL_F16)
    return tl_bn_train_stats_f16(x, ld, n, C, TL_BF16, gamma, beta, eps, momentum, ws, mean, rstd, scale, shift, running_mean, running_var, num_batches_tracked, stream);
#endif
  int blocks;
  const Part pt = partition(n, C, blocks);
  const size_t lds = (size_t)pt.rl * 2 * C * sizeof(double);
  hipStream_t s = tl_s(stream);
  if (dtype == TL_F32) k_bn_stats<float><<<blocks, kThreads, lds, s>>>((const float*)x, ld, n, C, pt, ws);
  else if (dtype == TL_BF16) k_bn_stats<__hip_bfloat16><<<blocks, kThreads, lds, s>>>((const __hip_bfloat16*)x, ld, n, C, pt, ws);
  else return TL_ERR_ARG;
  TL_CHECK_LAUNCH();
  k_bn_stats_finish<<<C, kFin, 0, s>>>(ws, blocks, n, C, gamma, beta, eps, momentum, mean, rstd, scale, shift, running_mean, running_var,
                                                 num_batches_tracked);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_bn_train_bwd(const void* x, int64_t ld, int x_dtype, const void* dy, int64_t dld, int dy_dtype, int64_t n, int C, const float* mean,
                    const float* rstd, const float* scale, const float* shift, int relu, double* ws, float* dgamma, float* dbeta, void* dx,
                    int64_t xld, const void* dx_add, int64_t ald, tl_stream_t stream) {
  if (!x || !dy || !mean || !rstd || !scale || !shift || !ws || !dgamma || !dbeta || !dx || n <= 0 || C <= 0 || C % 4 || C > 4 * kThreads || ld % 4 ||
      dld % 4 || xld % 4 || (dx_add && ald % 4))
    return TL_ERR_ARG;
#ifndef TL_F16_BUILD
  if (x_dtype == TL_F16 || dy_dtype == TL_F16) {
    if (x_dtype == TL_BF16 || dy_dtype == TL_BF16) return TL_ERR_ARG;
    return tl_bn_train_bwd_f16(x, ld, tl_f16_code(x_dtype), dy, dld, tl_f16_code(dy_dtype), n, C, mean, rstd, scale, shift, relu, ws, dgamma, dbeta, dx, xld, dx_add, ald,
                               stream);
  }
#endif
  int blocks;
  const Part pt = partition(n, C, blocks);
  const size_t lds = (size_t)pt.rl * 2 * C * sizeof(double);
  hipStream_t s = tl_s(stream);
  const unsigned g = tl_grid(n * (C / 4), kThreads);
#define TL_BN_BWD(TX, TG)                                                                                                              \
  do {                                                                                                                                 \
    k_bn_bwd_reduce<TX, TG><<<blocks, kThreads, lds, s>>>((const TX*)x, ld, (const TG*)dy, dld, n, C, pt, mean, rstd, scale, shift, relu, ws); \
    TL_CHECK_LAUNCH();                                                                                                                 \
    k_bn_bwd_finish<<<C, kFin, 0, s>>>(ws, blocks, C, dgamma, dbeta);                                                        \
    TL_CHECK_LAUNCH();                                                                                                                 \
    if (relu && launch_apply_v8(x, ld, x_dtype, dy, dld, dy_dtype, n, C, mean, rstd, scale, shift, true, dgamma, dbeta, dx, xld, dx_add, ald, s)) { TL_CHECK_LAUNCH(); break; } \
    if (!relu && launch_apply_v8(x, ld, x_dtype, dy, dld, dy_dtype, n, C, mean, rstd, scale, shift, false, dgamma, dbeta, dx, xld, dx_add, ald, s)) { TL_CHECK_LAUNCH(); break; } \
    k_bn_bwd_apply<TX, TG><<<g, kThreads, 0, s>>>((const TX*)x, ld, (const TG*)dy, dld, n, C, mean, rstd, scale, shift, relu, dgamma, dbeta, (TX*)dx, xld, (const TX*)dx_add, ald); \
    TL_CHECK_LAUNCH();                                                                                                                 \
  } while (0)
  if (x_dtype == TL_F32 && dy_dtype == TL_F32) TL_BN_BWD(float, float);
  else if (x_dtype == TL_F32 && dy_dtype == TL_BF16) TL_BN_BWD(float, __hip_bfloat16);
  else if (x_dtype == TL_BF16 && dy_dtype == TL_BF16) TL_BN_BWD(__hip_bfloat16, __hip_bfloat16);
  else if (x_dtype == TL_BF16 && dy_dtype == TL_F32) TL_BN_BWD(__hip_bfloat16, float);
  else return TL_ERR_ARG;
#undef TL_BN_BWD
  return TL_OK;
}

}  // extern "C"
