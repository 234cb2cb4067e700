// The output layers of the two point-wise heads in TRAINING: Linear(C, 2) and Linear(C, 3) over millions of points
// (reference tree_learn/model/blocks.py:8-26 `MLP`, tree_learn.py:43-46; forward + weight gradient -- the input gradient is a
// 2/3 -> C "conv" and runs on k_conv_tinycin).  Two or three output channels are no matrix-core shape: the scalar fallback
// (k_conv_generic, one thread per OUTPUT ELEMENT re-reading the row) took 0.68 ms per call and the fp32-MFMA weight gradient 0.74 ms
// for 0.25 GB of traffic.  Here one thread owns one ROW: four 16-B loads bring its C = 32 channels, the CO x C weights sit in LDS and
// are read as broadcasts, CO dot products per row; HBM-bound (0.25 GB per call).
//
//   tl_launch_conv_tinycout : out[o][j] = sum_c W[j][c] * x[o][c]                       (K = 1, no rulebook), Cout <= 8, Cin % 8 == 0
//   tl_launch_wgrad_tinycout: gW[j][c]  = sum_o gout[o][j] * x[o][c]                    fixed-order partial sums per workgroup, then
//                                                                                       tl_launch_wgrad_reduce: deterministic
#include "tl_conv_internal.h"
#include "tl_f16_train.h"

int tl_launch_wgrad_reduce(const float* ws, int64_t nparts, int64_t per, float* gw, hipStream_t s, int K = 1, int Cout = 0, int Cin = 0, int ref_layout = 0);   // tl_wgrad_dense.hip

namespace {

template <bool BF16>
static __device__ __forceinline__ void load8(const void* base, int64_t elem, float (&v)[8]) {
  if constexpr (BF16) {
    const u32x4 q = *reinterpret_cast<const u32x4*>((const uint16_t*)base + elem);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf16_lo(q[i]); v[2 * i + 1] = bf16_hi(q[i]); }
  } else {
    const f32x4 a = *reinterpret_cast<const f32x4*>((const float*)base + elem), b = *reinterpret_cast<const f32x4*>((const float*)base + elem + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[i + 4] = b[i]; }
  }
}

// OF32: the result is stored as fp32 whatever the input dtype (tl_linear_small_f32: the heads' outputs keep fp32 resolution under
// mixed precision -- a bf16 offset of 8-16 m would be quantised to 3-6 cm)
template <bool BF16, int CO, bool OF32 = false>
__global__ void __launch_bounds__(256) k_conv_tinycout(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) float wsh[];        // [CO][Cin]
  for (int e = threadIdx.x; e < CO * p.Cin; e += 256)
    wsh[e] = BF16 ? bf16_lo((uint32_t)((const uint16_t*)p.w)[e]) : ((const float*)p.w)[e];
  __syncthreads();
  const int nv = p.Cin >> 3;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < p.n_out; o += (int64_t)gridDim.x * 256) {
    float acc[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) acc[j] = 0.f;
    for (int v = 0; v < nv; ++v) {
      float x[8];
      load8<BF16>(p.in, o * p.in_ld + v * 8, x);
#pragma unroll
      for (int j = 0; j < CO; ++j) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(&wsh[j * p.Cin + v * 8]), w1 = *reinterpret_cast<const f32x4*>(&wsh[j * p.Cin + v * 8 + 4]);
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[j] = fmaf(x[q], w0[q], acc[j]); acc[j] = fmaf(x[q + 4], w1[q], acc[j]); }
      }
    }
#pragma unroll
    for (int j = 0; j < CO; ++j) epi_store1<BF16 && !OF32>(p.out, p.out_ld, p.out_scale, p.out_shift, p.out_relu, o, j, acc[j]);
  }
}

// workgroup b owns rows [b * rpb, (b + 1) * rpb); thread t: channel group t % (Cin / 8), row lane t / (Cin / 8); the row lanes' sums are
// added in lane order through LDS -> one partial [CO][Cin] per workgroup
template <bool BF16, int CO>
__global__ void __launch_bounds__(256) k_wgrad_tinycout(const void* __restrict__ x, int64_t x_ld, const void* __restrict__ g, int64_t g_ld, int64_t n, int Cin,
                                                        int64_t rpb, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float red[];        // [rl][CO][Cin]
  const int cgs = Cin >> 3, rl = 256 / cgs;
  const int t = threadIdx.x, cg = t % cgs, lane = t / cgs;
  float acc[CO][8];
#pragma unroll
  for (int j = 0; j < CO; ++j)
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[j][q] = 0.f;
  if (lane < rl) {
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = min(n, r0 + rpb);
#pragma unroll 2
    for (int64_t r = r0 + lane; r < r1; r += rl) {
      float xv[8], gv[CO];
      load8<BF16>(x, r * x_ld + cg * 8, xv);
#pragma unroll
      for (int j = 0; j < CO; ++j) gv[j] = BF16 ? bf16_lo((uint32_t)((const uint16_t*)g)[r * g_ld + j]) : ((const float*)g)[r * g_ld + j];
#pragma unroll
      for (int j = 0; j < CO; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[j][q] = fmaf(gv[j], xv[q], acc[j][q]);
    }
#pragma unroll
    for (int j = 0; j < CO; ++j)
#pragma unroll
      for (int q = 0; q < 8; ++q) red[((int64_t)lane * CO + j) * Cin + cg * 8 + q] = acc[j][q];
  }
  __syncthreads();
  for (int e = t; e < CO * Cin; e += 256) {
    float s = 0.f;
    for (int l = 0; l < rl; ++l) s += red[(int64_t)l * CO * Cin + e];
    part[(int64_t)blockIdx.x * CO * Cin + e] = s;
  }
}

}  // namespace

int tl_launch_conv_tinycout(const ConvP& p, int dtype, hipStream_t s) {
  if (p.K != 1 || p.table || p.Cout < 1 || p.Cout > 8 || p.Cin % 8 || p.Cin > 1024 || p.res || p.out2 || p.out3 || p.in_scale || p.in_relu) return TL_ERR_UNSUPPORTED;
  const int eb = dtype == TL_BF16 ? 2 : 4;
  if (p.in_ld % 8 || ((uintptr_t)p.in) % 16 || ((uintptr_t)p.w) % eb) return TL_ERR_UNSUPPORTED;
  const unsigned g = tl_grid(p.n_out, 256);
  const size_t lds = (size_t)p.Cout * p.Cin * 4;
#define TL_TC(CO_)                                                                     \
  case CO_:                                                                            \
    if (dtype == TL_BF16) k_conv_tinycout<true, CO_><<<g, 256, lds, s>>>(p);           \
    else k_conv_tinycout<false, CO_><<<g, 256, lds, s>>>(p);                           \
    break;
  switch (p.Cout) { TL_TC(1) TL_TC(2) TL_TC(3) TL_TC(4) TL_TC(5) TL_TC(6) TL_TC(7) TL_TC(8) }
#undef TL_TC
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

// partial tile sets (of Cout * Cin floats) the kernel writes for n rows
int64_t tl_wgrad_tinycout_parts(int64_t n) {
  int64_t b = tl_cdiv(n, 2048);
  if (b > 1024) b = 1024;
  return b < 1 ? 1 : b;
}

int tl_launch_wgrad_tinycout(const void* x, int64_t x_ld, const void* g, int64_t g_ld, int dtype, int64_t n, int Cin, int Cout, float* gw, float* ws, hipStream_t s) {
  if (Cout < 1 || Cout > 4 || Cin % 8 || Cin > 256 || 256 % (Cin / 8) || (Cout * Cin) % 4 || x_ld % 8 || ((uintptr_t)x) % 16 || ((uintptr_t)ws) % 16 || ((uintptr_t)gw) % 16)
    return TL_ERR_UNSUPPORTED;
  const int64_t blocks = tl_wgrad_tinycout_parts(n), rpb = tl_cdiv(n, blocks);
  const int64_t used = tl_cdiv(n, rpb);
  const size_t lds = (size_t)(256 / (Cin / 8)) * Cout * Cin * 4;
  if (lds > 64 * 1024) return TL_ERR_UNSUPPORTED;
#define TL_TW(CO_)                                                                                                            \
  case CO_:                                                                                                                   \
    if (dtype == TL_BF16) k_wgrad_tinycout<true, CO_><<<(unsigned)used, 256, lds, s>>>(x, x_ld, g, g_ld, n, Cin, rpb, ws);     \
    else k_wgrad_tinycout<false, CO_><<<(unsigned)used, 256, lds, s>>>(x, x_ld, g, g_ld, n, Cin, rpb, ws);                     \
    break;
  switch (Cout) { TL_TW(1) TL_TW(2) TL_TW(3) TL_TW(4) }
#undef TL_TW
  if (hipGetLastError() != hipSuccess) return TL_ERR_LAUNCH;
  return tl_launch_wgrad_reduce(ws, used, (int64_t)Cout * Cin, gw, s);
}

extern "C" int tl_linear_small_f32(const void* x, int64_t x_ld, int dtype, const void* w, int Cin, int Cout, int64_t n, float* out, int64_t out_ld,
                                   tl_stream_t stream) {
#ifndef TL_F16_BUILD
  if (dtype == TL_F16) return tl_linear_small_f32_f16(x, x_ld, TL_BF16, w, Cin, Cout, n, out, out_ld, stream);
#endif
  if (!x || !w || !out || n <= 0 || Cin <= 0 || Cout < 1 || Cout > 8 || (dtype != TL_F32 && dtype != TL_BF16)) return TL_ERR_ARG;
  if (Cin % 8 || Cin > 1024 || x_ld % 8 || ((uintptr_t)x) % 16) return TL_ERR_UNSUPPORTED;
  ConvP p{};
  p.in = x; p.in_ld = x_ld; p.w = w; p.n_out = n; p.n_in = n; p.K = 1; p.Cin = Cin; p.Cout = Cout; p.out = out; p.out_ld = out_ld;
  const unsigned g = tl_grid(n, 256);
  const size_t lds = (size_t)Cout * Cin * 4;
  hipStream_t s = tl_s(stream);
#define TL_TF(CO_)                                                                             \
  case CO_:                                                                                    \
    if (dtype == TL_BF16) k_conv_tinycout<true, CO_, true><<<g, 256, lds, s>>>(p);             \
    else k_conv_tinycout<false, CO_, true><<<g, 256, lds, s>>>(p);                             \
    break;
  switch (Cout) { TL_TF(1) TL_TF(2) TL_TF(3) TL_TF(4) TL_TF(5) TL_TF(6) TL_TF(7) TL_TF(8) }
#undef TL_TF
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}
