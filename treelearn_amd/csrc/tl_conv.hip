// Sparse convolution forward (SubMConv3d / SparseConv3d / SparseInverseConv3d / 1x1) for gfx950.
//
// Output-stationary gather-GEMM: a workgroup owns 128 consecutive output rows (ascending voxel
// key => spatially adjacent), walks the kernel taps that have at least one present neighbour in the
// tile, gathers the neighbour rows (16-B loads, 128 B contiguous per row segment), applies the
// fused BatchNorm+ReLU prologue, stages them with the tap's weight slice in padded LDS and
// contracts on the matrix cores; every output row is written exactly once (no atomics =>
// bit-deterministic), with the residual add / BatchNorm+ReLU epilogue fused into the store.
//
//   fp32 : v_mfma_f32_32x32x2_f32 (exact fp32, bitwise an fmaf chain) -- the parity path.
//   bf16 : v_mfma_f32_32x32x16_bf16, fp32 accumulate                  -- the throughput path.
//
// LDS row pitch = 32 channels + 16 B pad: ds_read_b128 of 16 different rows at one column offset
// lands on 16 different 16-B slots (36*r mod 64 is a permutation of multiples of 4 for r mod 16),
// i.e. conflict-free for the b128 lane groups of MI355X_MICROARCH.md §LDS.
#include "tl_conv_internal.h"
#include <atomic>
#include <string.h>

namespace {

// ------------------------------------------------------------------ generic fallback (any Cin/Cout)
template <typename T>
__global__ void __launch_bounds__(256) k_conv_generic(ConvP p) {
  const T* in = (const T*)p.in; const T* w = (const T*)p.w; const T* res = (const T*)p.res;
  const int64_t total = p.n_out * p.Cout;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = t / p.Cout; const int j = (int)(t % p.Cout);
    float acc = 0.f;
    for (int k = 0; k < p.K; ++k) {
      const int64_t idx = p.table ? (int64_t)p.table[(int64_t)k * p.n_out + o] : o;
      if (idx < 0) continue;
      const T* x = in + idx * p.in_ld;
      const T* wr = w + ((int64_t)k * p.Cout + j) * p.Cin;
      for (int c = 0; c < p.Cin; ++c) {
        float v = ld_elem(x + c);
        if (p.in_scale) v = fmaf(v, p.in_scale[c], p.in_shift[c]);
        if (p.in_relu) v = fmaxf(v, 0.f);
        if (sizeof(T) == 2) v = __bfloat162float(__float2bfloat16(v));   // bf16 path feeds bf16 to the MFMA
        acc = fmaf(v, ld_elem(wr + c), acc);
      }
    }
    if (res) acc += ld_elem(res + o * p.res_ld + j);
    constexpr bool BF = sizeof(T) == 2;
    epi_store1<BF>(p.out, p.out_ld, p.out_scale, p.out_shift, p.out_relu, o, j, acc);
    if (p.out2) epi_store1<BF>(p.out2, p.out2_ld, p.out2_scale, p.out2_shift, p.out2_relu, o, j, acc);
    if (p.out3) epi_store1<BF>(p.out3, p.out3_ld, p.out3_scale, p.out3_shift, p.out3_relu, o, j, acc);
  }
}

// ------------------------------------------------------------------ fp32 MFMA kernel
constexpr int TM = 128;          // output rows per workgroup
constexpr int64_t kSmallRows = 16384; // below this the 128-row tiling cannot fill 256 CUs: use tl_conv_small
constexpr int KC = 32;           // input channels per step
constexpr int LDA = KC + 4;      // padded LDS pitch in floats (144 B)

template <int NB>                // NB = Cout / 32
__global__ void __launch_bounds__(256) k_conv_mfma_f32(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* idx_s = reinterpret_cast<int*>(smem);                                   // [K][TM]
  unsigned* mask_s = reinterpret_cast<unsigned*>(idx_s + p.K * TM);             // [4] (16 B keeps alignment)
  float* As = reinterpret_cast<float*>(mask_s + 4);                             // [2][TM][LDA]
  float* Bs = As + 2 * TM * LDA;                                                // [2][NB*32][LDA]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tile = xcd_tile(blockIdx.x, p.nblk);
  const int64_t r0 = (int64_t)tile * TM;
  const float* in = (const float*)p.in; const float* W = (const float*)p.w;

  if (tid == 0) mask_s[0] = 0;
  __syncthreads();
  unsigned local = 0;
  for (int e = tid; e < p.K * TM; e += 256) {
    const int k = e >> 7, r = e & (TM - 1);
    const int64_t row = r0 + r;
    int idx = -1;
    if (row < p.n_out) idx = p.table ? p.table[(int64_t)k * p.n_out + row] : (int)row;
    idx_s[e] = idx;
    if (idx >= 0) local |= 1u << k;
  }
  if (local) atomicOr(&mask_s[0], local);
  __syncthreads();
  unsigned mask = __builtin_amdgcn_readfirstlane(mask_s[0]);

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

  const int nchunk = p.Cin / KC;
  const int lrow = tid >> 3, c4 = (tid & 7) * 4;
  float4 ra[4];
  f32x4 rb[NB];                      // native vector type: HIP's float4 struct copies lower to memcpy and pin the array in scratch
  bool va[4];

  auto issue = [&](int k, int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = idx_s[k * TM + lrow + 32 * i];
      va[i] = idx >= 0;
      ra[i] = va[i] ? *reinterpret_cast<const float4*>(in + (int64_t)idx * p.in_ld + ch * KC + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      rb[i] = *reinterpret_cast<const f32x4*>(W + ((int64_t)k * p.Cout + lrow + 32 * i) * p.Cin + ch * KC + c4);
  };
  auto stage = [&](int buf, int ch) __attribute__((always_inline)) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.in_scale) {
      sc = *reinterpret_cast<const float4*>(p.in_scale + ch * KC + c4);
      sh = *reinterpret_cast<const float4*>(p.in_shift + ch * KC + c4);
    }
    float* a = As + buf * TM * LDA;
    float* b = Bs + buf * NB * 32 * LDA;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = ra[i];
      if (va[i]) {
        if (p.in_scale) { v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w); }
        if (p.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      *reinterpret_cast<float4*>(a + (lrow + 32 * i) * LDA + c4) = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(b + (lrow + 32 * i) * LDA + c4) = rb[i];
  };

  if (mask) {
    int k = __builtin_ctz(mask), ch = 0;
    issue(k, ch);
    stage(0, ch);
    __syncthreads();
    int buf = 0;
    const int fi = lane & 31, fh = lane >> 5;
    while (true) {
      // next step
      int nk = k, nch = ch + 1;
      unsigned nmask = mask;
      if (nch == nchunk) { nch = 0; nmask = mask & (mask - 1); nk = nmask ? __builtin_ctz(nmask) : -1; }
      const bool more = nk >= 0;
      if (more) issue(nk, nch);

      const float* a = As + buf * TM * LDA + (wv * 32 + fi) * LDA + fh * 4;
      float4 af[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const float4*>(a + 8 * j);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float* b = Bs + buf * NB * 32 * LDA + (nb * 32 + fi) * LDA + fh * 4;
        float4 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const float4*>(b + 8 * j);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j].x, bf[j].x, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j].y, bf[j].y, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j].z, bf[j].z, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j].w, bf[j].w, acc[nb], 0, 0, 0);
        }
      }
      if (!more) break;
      stage(buf ^ 1, nch);
      __syncthreads();
      buf ^= 1; k = nk; ch = nch; mask = nmask;
    }
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const float* res = (const float*)p.res;
  const int col = lane & 31;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int j = nb * 32 + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = r0 + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row >= p.n_out) continue;
      float v = acc[nb][r];
      if (res) v += res[row * p.res_ld + j];
      epi_store1<false>(p.out, p.out_ld, p.out_scale, p.out_shift, p.out_relu, row, j, v);
      if (p.out2) epi_store1<false>(p.out2, p.out2_ld, p.out2_scale, p.out2_shift, p.out2_relu, row, j, v);
      if (p.out3) epi_store1<false>(p.out3, p.out3_ld, p.out3_scale, p.out3_shift, p.out3_relu, row, j, v);
    }
  }
}

template <int NB>
int launch_mfma_f32(const ConvP& p, hipStream_t s) {
  const size_t lds = (size_t)p.K * TM * 4 + 16 + 2 * (size_t)TM * LDA * 4 + 2 * (size_t)NB * 32 * LDA * 4;
  static TlAttrOnce attr_once;                     // per kernel instantiation AND device (the attribute is per device)
  if (!tl_lds_attr(attr_once, reinterpret_cast<const void*>(&k_conv_mfma_f32<NB>), 160 * 1024)) return TL_ERR_LAUNCH;
  k_conv_mfma_f32<NB><<<p.nblk, 256, lds, s>>>(p);
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

__global__ void k_pack_weight(const float* __restrict__ w, int Cout, int K, int Cin, void* __restrict__ o, int dtype) {
  const int64_t total = (int64_t)Cout * K * Cin;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % Cin); const int64_t r = t / Cin; const int j = (int)(r % Cout); const int k = (int)(r / Cout);
    const float v = w[((int64_t)j * K + k) * Cin + c];                 // reference layout [Cout][K][Cin]
    if (dtype == TL_F32) reinterpret_cast<float*>(o)[t] = v;
    else if (dtype == TL_F16) reinterpret_cast<_Float16*>(o)[t] = (_Float16)v;
    else reinterpret_cast<__hip_bfloat16*>(o)[t] = __float2bfloat16(v);
  }
}

// split-bf16 form of fp32 weights (tl_conv_internal.h: the bf16x3 contraction): [K][Cout][Cin / 32][64 x bf16] -- per 32-channel unit the hi
// parts of the four 8-channel pieces (J, fh) at 16-B slots 2 J + fh, the lo parts at slots 4 + 2 J + fh; piece (J, fh) = channels
// 16 J + 4 fh + {0..3}, 16 J + 8 + 4 fh + {0..3} of the unit
__global__ void k_pack_weight_x3(const float* __restrict__ w, int Cout, int K, int Cin, uint16_t* __restrict__ o) {
  // Cin >= 256: stored as TWO independent half-width convs (input channels [0, Cin / 2) then [Cin / 2, Cin)), each [K][Cout][Cin / 64][64]:
  // tl_conv_fwd runs such a conv as two launches of the kernel that has an instantiation for the half width
  const int un_all = Cin / 32, halves = Cin >= 256 ? 2 : 1, un = un_all / halves;
  const int64_t per_half = (int64_t)Cout * K * un * 64, total = per_half * halves;
  for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t0 < total; t0 += (int64_t)gridDim.x * blockDim.x) {
    const int hf = (int)(t0 / per_half); const int64_t t = t0 - hf * per_half;
    const int e = (int)(t & 63); const int64_t r = t >> 6; const int c = (int)(r % un); const int64_t r2 = r / un;
    const int n = (int)(r2 % Cout); const int k = (int)(r2 / Cout);
    const int half = e >> 5, piece = (e & 31) >> 3, q = e & 7, J = piece >> 1, fh = piece & 1;
    const int ch = 32 * (c + hf * un) + 16 * J + (q < 4 ? 4 * fh + q : 8 + 4 * fh + (q - 4));
    const float v = w[((int64_t)n * K + k) * Cin + ch];                // reference layout [Cout][K][Cin]
    const __hip_bfloat16 hi = __float2bfloat16(v);
    const __hip_bfloat16 lo = __float2bfloat16(v - __bfloat162float(hi));
    o[t0] = half ? __builtin_bit_cast(uint16_t, lo) : __builtin_bit_cast(uint16_t, hi);
    // second copy in MFMA-fragment order (Cout % 32 == 0, Cin < 256) behind the first: the 16-B piece j (0, 1 = hi of channel group J = j; 2, 3 =
    // lo of J = j - 2) of lane (n & 31, fh) for (tap, column block, unit) is one contiguous KB -- what the small-level kernel loads per
    // instruction (tl_conv_small.hip, FR): slot 2 J + fh / 4 + 2 J + fh of the record above
    if (halves == 1 && Cout % 32 == 0) {
      const int j = half ? 2 + J : J;
      const int lane = (n & 31) + 32 * fh, cbt = n >> 5, CBt = Cout >> 5;
      o[total + ((((((int64_t)k * CBt + cbt) * un + c) * 4 + j) * 64 + lane) * 8) + q] = half ? __builtin_bit_cast(uint16_t, lo) : __builtin_bit_cast(uint16_t, hi);
    }
  }
}

// weights of the input-gradient conv straight from the reference layout: o[k][ci][co] = w[co][flip ? K-1-k : k][ci]
__global__ void k_pack_weight_dgrad(const float* __restrict__ w, int Cout, int K, int Cin, int flip, void* __restrict__ o, int dtype) {
  const int64_t total = (int64_t)Cout * K * Cin;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(t % Cout); const int64_t r = t / Cout; const int ci = (int)(r % Cin); const int k = (int)(r / Cin);
    const float v = w[((int64_t)co * K + (flip ? K - 1 - k : k)) * Cin + ci];
    if (dtype == TL_F32) reinterpret_cast<float*>(o)[t] = v;
    else if (dtype == TL_F16) reinterpret_cast<_Float16*>(o)[t] = (_Float16)v;
    else reinterpret_cast<__hip_bfloat16*>(o)[t] = __float2bfloat16(v);
  }
}

// fragment order: vector v = ((((k*CB + cb)*CH + ch)*J + j)*64 + lane) holds 16 B = W[k][32cb + (lane&31)][32ch + (32j + 16(lane>>5))/EB ..]
__global__ void k_pack_weight_frag(const float* __restrict__ w, int Cout, int K, int Cin, void* __restrict__ o, int dtype) {
  const int EB = dtype == TL_F32 ? 4 : 2, J = 32 * EB / 32, EPV = 16 / EB;
  const int CB = Cout / 32, CH = Cin / 32;
  const int64_t nvec = (int64_t)K * CB * CH * J * 64;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(v & 63); int64_t r = v >> 6;
    const int j = (int)(r % J); r /= J; const int ch = (int)(r % CH); r /= CH; const int cb = (int)(r % CB); const int k = (int)(r / CB);
    const int col = cb * 32 + (lane & 31), c0 = ch * 32 + (32 * j + 16 * (lane >> 5)) / EB;
    for (int e = 0; e < EPV; ++e) {
      const float x = w[((int64_t)col * K + k) * Cin + c0 + e];
      if (dtype == TL_F32) reinterpret_cast<float*>(o)[v * EPV + e] = x;
      else if (dtype == TL_F16) reinterpret_cast<_Float16*>(o)[v * EPV + e] = (_Float16)x;
      else reinterpret_cast<__hip_bfloat16*>(o)[v * EPV + e] = __float2bfloat16(x);
    }
  }
}

// Every conv weight of a model in ONE launch (training: the optimizer changes all of them each step; per-layer packing was ~210 launches).
// Workgroup b serves 4096 consecutive output elements of ONE descriptor (blocks[b] = {descriptor, first element}); forms: 0 = [K][Cout][Cin],
// 1 = fragment order (k_pack_weight_frag), 2 / 3 = the input-gradient layout [K][Cin][Cout] without / with flipped taps.
__global__ void __launch_bounds__(256) k_pack_batch(const tl_pack_desc* __restrict__ descs, const int2* __restrict__ blocks, int dtype) {
  const int2 bd = blocks[blockIdx.x];
  const tl_pack_desc d = descs[bd.x];
  const int64_t total = (int64_t)d.Cout * d.K * d.Cin;
  const int EB = dtype == TL_F32 ? 4 : 2, EPV = 16 / EB;
  for (int i = 0; i < 16; ++i) {
    const int64_t t = (int64_t)bd.y * 4096 + i * 256 + threadIdx.x;
    if (t >= total) break;
    float v;
    if (d.form == 0) {
      const int c = (int)(t % d.Cin); const int64_t r = t / d.Cin; const int j = (int)(r % d.Cout); const int k = (int)(r / d.Cout);
      v = d.src[((int64_t)j * d.K + k) * d.Cin + c];
    } else if (d.form == 1) {
      const int J = 32 * EB / 32, CB = d.Cout / 32, CH = d.Cin / 32;
      const int64_t vec = t / EPV; const int e = (int)(t % EPV);
      const int lane = (int)(vec & 63); int64_t r = vec >> 6;
      const int j = (int)(r % J); r /= J; const int ch = (int)(r % CH); r /= CH; const int cb = (int)(r % CB); const int k = (int)(r / CB);
      const int col = cb * 32 + (lane & 31), c0 = ch * 32 + (32 * j + 16 * (lane >> 5)) / EB;
      v = d.src[((int64_t)col * d.K + k) * d.Cin + c0 + e];
    } else {
      const int co = (int)(t % d.Cout); const int64_t r = t / d.Cout; const int ci = (int)(r % d.Cin); const int k = (int)(r / d.Cin);
      v = d.src[((int64_t)co * d.K + (d.form == 3 ? d.K - 1 - k : k)) * d.Cin + ci];
    }
    if (dtype == TL_F32) reinterpret_cast<float*>(d.dst)[t] = v;
    else if (dtype == TL_F16) reinterpret_cast<_Float16*>(d.dst)[t] = (_Float16)v;
    else reinterpret_cast<__hip_bfloat16*>(d.dst)[t] = __float2bfloat16(v);
  }
}

template <typename T>
__global__ void k_affine_relu(const T* __restrict__ in, int64_t in_ld, T* __restrict__ out, int64_t out_ld, int64_t n, int C,
                              const float* __restrict__ scale, const float* __restrict__ shift, int relu) {
  const int64_t total = n * C;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / C; const int c = (int)(t % C);
    float v = ld_elem(in + r * in_ld + c);
    if (scale) v = fmaf(v, scale[c], shift[c]);
    if (relu) v = fmaxf(v, 0.f);
    st_elem(out + r * out_ld + c, v);
  }
}

// The same over 8-channel vectors (C % 8 == 0, 16-B aligned rows): one 16-B (bf16) / two 16-B (f32) loads and stores per thread and
// step; the grid stride is a multiple of the vectors per row, so a thread keeps ONE channel group and its scale / shift live in
// registers (the scalar kernel above re-reads them per element and moves 2 bytes per load: 1.7 TB/s on the training step's 67 passes).
// T16: 0 = fp32 rows, 1 = bf16, 2 = IEEE half (this unit is compiled once: the half conversions are spelled out here)
template <int T16>
__global__ void __launch_bounds__(256) k_affine_relu_v8(const void* __restrict__ in, int64_t in_ld, void* __restrict__ out, int64_t out_ld, int64_t n, int C,
                                                        const float* __restrict__ scale, const float* __restrict__ shift, int relu) {
  const int vpr = C >> 3;
  const int64_t total = n * vpr, stride = (int64_t)gridDim.x * 256;
  int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= total) return;
  const int c0 = (int)(v % vpr) * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { sc[q] = scale ? scale[c0 + q] : 1.f; sh[q] = scale ? shift[c0 + q] : 0.f; }
  for (; v < total; v += stride) {
    const int64_t r = v / vpr;
    float x[8];
    typedef _Float16 hx2 __attribute__((ext_vector_type(2)));
    if constexpr (T16 == 1) {
      const u32x4 q4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>((const uint16_t*)in + r * in_ld + c0));
#pragma unroll
      for (int q = 0; q < 4; ++q) { x[2 * q] = bf16_lo(q4[q]); x[2 * q + 1] = bf16_hi(q4[q]); }
    } else if constexpr (T16 == 2) {
      // (a plain 16-B load: with __builtin_nontemporal_load hipcc 7.0 fetched ONE dword here and fed all eight v_fma_mix from it)
      const u32x4 q4 = *reinterpret_cast<const u32x4*>((const uint16_t*)in + r * in_ld + c0);
      const uint32_t w0 = q4[0], w1 = q4[1], w2 = q4[2], w3 = q4[3];
      const uint32_t ws[4] = {w0, w1, w2, w3};
#pragma unroll
      for (int q = 0; q < 4; ++q) { const hx2 h = __builtin_bit_cast(hx2, ws[q]); x[2 * q] = (float)h[0]; x[2 * q + 1] = (float)h[1]; }
    } else {
      const f32x4* s4 = reinterpret_cast<const f32x4*>((const float*)in + r * in_ld + c0);
      const f32x4 a = __builtin_nontemporal_load(s4), b = __builtin_nontemporal_load(s4 + 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) { x[q] = a[q]; x[q + 4] = b[q]; }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { x[q] = fmaf(x[q], sc[q], sh[q]); if (relu) x[q] = fmaxf(x[q], 0.f); }
    if constexpr (T16 == 1) {
      u32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = pack_bf16x2(x[2 * q], x[2 * q + 1]);
      *reinterpret_cast<u32x4*>((uint16_t*)out + r * out_ld + c0) = o;
    } else if constexpr (T16 == 2) {
      u32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const hx2 h = {(_Float16)x[2 * q], (_Float16)x[2 * q + 1]}; o[q] = __builtin_bit_cast(uint32_t, h); }
      *reinterpret_cast<u32x4*>((uint16_t*)out + r * out_ld + c0) = o;
    } else {
      f32x4* d = reinterpret_cast<f32x4*>((float*)out + r * out_ld + c0);
      d[0] = f32x4{x[0], x[1], x[2], x[3]}; d[1] = f32x4{x[4], x[5], x[6], x[7]};
    }
  }
}

}  // namespace

static int64_t g_small_rows = kSmallRows;
static int g_dbg = 0;
static int g_bf16_depth = 0, g_bf16_units = 0;   // 0 = kernel default
extern int g_small_mode;                          // tl_conv_small.hip
extern int g_head_mode;                           // tl_head.hip
extern int g_wgrad_dense;                         // tl_wgrad_dense.hip
extern int64_t g_wgrad_dense_min_rows;
extern int g_wgrad_rows;                          // tl_wgrad_rows.hip
extern int g_wgrad_dma, g_wgrad_dense_gx;         // tl_wgrad_dense.hip
static int g_stream = 1;                          // use the streamed-weights register-gather kernel where it applies
static int g_streamq = 1;                         // ... and its quad-gather form for bf16 with Cin % 64 == 0
static int g_streamq_x3 = 1;                      // ... and for fp32 rows in the parity-fast mode (tl_set_tuning "streamq_x3")
static int g_direct = 1;                          // use the weights-in-LDS direct kernel where it applies
static int g_direct_oh = 1;                       // ... and its gather-once form for the level-1 inverse conv
static int g_blk = 1;                             // use the staged-unit kernel when the caller passes the block-local rulebook form
static int g_up = 1;                              // use the coarse-stationary inverse conv when the caller passes the scatter form of the table
#ifdef TL_DEV                                     // the window kernel lives in the developer build only (python -m treelearn_amd.build --dev)
static int g_win = 0;                             // window kernel (opt-in, TL_CONV_WIN=1: measured at parity with the gather kernels): 1 = shapes with >= 64 channels, 2 = all, 0 = off
extern int g_win_rows, g_win_ct;                  // tl_conv_win.hip
static int64_t g_win_min_rows = 65536;            // below this a 512-row tiling leaves most CUs idle
#endif

// (tl_exec.hip: may the forward hand the level-1 inverse conv its packed table?  Developer switches can take the gather-once kernel away.)
bool tl_conv_one_hot_direct_enabled(int64_t n_out) { return g_direct && g_direct_oh && n_out > g_small_rows; }     // (small levels: the small-level kernel serves the conv, from the [8][n] table)

extern "C" {

int tl_set_tuning(const char* key, int64_t value) {
  if (!key) return TL_ERR_ARG;
  if (!strcmp(key, "bf16_depth")) { g_bf16_depth = (int)value; return TL_OK; }
  if (!strcmp(key, "bf16_units")) { g_bf16_units = (int)value; return TL_OK; }
  if (!strcmp(key, "direct")) { g_direct = (int)value; return TL_OK; }
  if (!strcmp(key, "blk")) { g_blk = (int)value; return TL_OK; }
  if (!strcmp(key, "up")) { g_up = (int)value; return TL_OK; }
  if (!strcmp(key, "direct_oh")) { g_direct_oh = (int)value; return TL_OK; }
#ifdef TL_DEV
  if (!strcmp(key, "win")) { g_win = (int)value; return TL_OK; }
  if (!strcmp(key, "win_rows")) { g_win_rows = (int)value; return TL_OK; }
  if (!strcmp(key, "win_ct")) { g_win_ct = (int)value; return TL_OK; }
  if (!strcmp(key, "win_min_rows")) { g_win_min_rows = value; return TL_OK; }
#else
  if (!strncmp(key, "win", 3)) return TL_ERR_UNSUPPORTED;         // developer build only
#endif
  if (!strcmp(key, "stream")) { g_stream = (int)value; return TL_OK; }
  if (!strcmp(key, "streamq")) { g_streamq = (int)value; return TL_OK; }
  if (!strcmp(key, "streamq_x3")) { g_streamq_x3 = (int)value; return TL_OK; }
  if (!strcmp(key, "stream_rb")) return tl_stream_set_rb((int)value);
  if (!strcmp(key, "x3_chunks")) return tl_conv_blk_x3_set_chunks((int)value);
  if (!strcmp(key, "small_rows")) { g_small_rows = value; return TL_OK; }
  if (!strcmp(key, "small_mode")) { g_small_mode = (int)value; return TL_OK; }
  if (!strcmp(key, "head_mode")) { g_head_mode = (int)value; return TL_OK; }
  if (!strcmp(key, "dbg")) { g_dbg = (int)value; return TL_OK; }
  if (!strcmp(key, "wgrad_dense")) { g_wgrad_dense = (int)value; return TL_OK; }
  if (!strcmp(key, "wgrad_rows")) { g_wgrad_rows = (int)value; return TL_OK; }
  if (!strcmp(key, "wgrad_dma")) { g_wgrad_dma = (int)value; return TL_OK; }
  if (!strcmp(key, "wgrad_dense_gx")) { g_wgrad_dense_gx = (int)value; return TL_OK; }
  if (!strcmp(key, "wgrad_dense_min_rows")) { g_wgrad_dense_min_rows = value; return TL_OK; }
  return TL_ERR_ARG;
}

int tl_conv_fwd(const tl_conv_args* a, tl_stream_t stream) {
  if (!a || !a->in || !a->weight || !a->out || a->n_out <= 0 || a->K <= 0 || a->K > 27 || a->Cin <= 0 || a->Cout <= 0) return TL_ERR_ARG;
  const bool has_blk = a->blk_unit && a->blk_counter && a->blk_halo && a->blk_lrb && a->K == 27;
  if (!a->table && a->K != 1 && !has_blk && !(a->in_all_ones && a->blk_pmask)) return TL_ERR_ARG;
  if ((a->in_scale == nullptr) != (a->in_shift == nullptr)) return TL_ERR_ARG;
  if ((a->out_scale == nullptr) != (a->out_shift == nullptr)) return TL_ERR_ARG;
  if (a->dtype != TL_F32 && a->dtype != TL_BF16 && a->dtype != TL_F16) return TL_ERR_ARG;
  // float16: the same kernel sources compiled a second time with IEEE-half conversions (tl_half.h); inside those units -- and in the
  // dispatch below -- "TL_BF16" stands for "the 16-bit type"
  const bool f16 = a->dtype == TL_F16;
  const int dt = f16 ? TL_BF16 : a->dtype;
  const auto L_direct = f16 ? tl_launch_conv_direct_f16 : tl_launch_conv_direct;
  const auto L_ones27 = f16 ? tl_launch_conv_ones27_f16 : tl_launch_conv_ones27;
  const auto L_stream = f16 ? tl_launch_conv_stream_f16 : tl_launch_conv_stream;
  const auto L_streamq = f16 ? tl_launch_conv_streamq_f16 : tl_launch_conv_streamq;
  const auto L_small = f16 ? tl_launch_conv_small_f16 : tl_launch_conv_small;
  const auto L_tinycin = f16 ? tl_launch_conv_tinycin_f16 : tl_launch_conv_tinycin;
  const auto L_blk = f16 ? tl_launch_conv_blk_f16 : tl_launch_conv_blk;
  const auto L_up = f16 ? tl_launch_conv_up_f16 : tl_launch_conv_up;
  ConvP p;
  p.in = a->in; p.in_ld = a->in_ld; p.w = a->weight; p.w_frag = a->weight_frag;
  p.w_x3 = (a->dtype == TL_F32 && a->Cin % 32 == 0 && ((uintptr_t)a->weight_x3) % 16 == 0) ? a->weight_x3 : nullptr; p.table = a->table; p.ctab = (a->K == 27) ? a->table_compact : nullptr; p.n_out = a->n_out; p.n_in = a->n_in;
  p.K = a->K; p.Cin = a->Cin; p.Cout = a->Cout; p.in_scale = a->in_scale; p.in_shift = a->in_shift;
  p.in_relu = a->in_relu; p.out_relu = a->out_relu; p.res = a->residual; p.res_ld = a->res_ld;
  p.out_scale = a->out_scale; p.out_shift = a->out_shift; p.out = a->out; p.out_ld = a->out_ld;
  p.out2 = a->out2; p.out2_ld = a->out2_ld; p.out2_scale = a->out2_scale; p.out2_shift = a->out2_shift; p.out2_relu = a->out2_relu;
  p.out3 = a->out3; p.out3_ld = a->out3_ld; p.out3_scale = a->out3_scale; p.out3_shift = a->out3_shift; p.out3_relu = a->out3_relu;
  if ((a->out2_scale == nullptr) != (a->out2_shift == nullptr) || (a->out3_scale == nullptr) != (a->out3_shift == nullptr)) return TL_ERR_ARG;
  p.nblk = (int)tl_cdiv(a->n_out, TM);
  p.dbg = g_dbg; p.one_hot = a->table != nullptr ? a->table_one_hot : 0;
  p.blk_unit = has_blk ? a->blk_unit : nullptr; p.blk_counter = a->blk_counter; p.blk_halo = a->blk_halo; p.blk_lrb = a->blk_lrb;
  p.blk_pmask = a->K == 27 ? a->blk_pmask : nullptr;
  p.epi_mode = a->epi_mode; p.red_part = a->red_part; p.red_nparts = a->red_nparts; p.bn_x = a->bn_x; p.bn_x_ld = a->bn_x_ld;
  p.bn_mean = a->bn_mean; p.bn_rstd = a->bn_rstd; p.bn_scale = a->bn_scale; p.bn_shift = a->bn_shift; p.bn_relu = a->bn_relu;
  const bool train = a->epi_mode != TL_EPI_NONE;             // only the direct / stream families carry the training epilogues
  if (train) {
    if ((a->epi_mode != TL_EPI_STATS && a->epi_mode != TL_EPI_BN_BWD) || !a->red_part || ((uintptr_t)a->red_part) % 8) return TL_ERR_ARG;
    if (a->epi_mode == TL_EPI_BN_BWD) {
      if (!a->bn_x || !a->bn_mean || !a->bn_rstd || !a->bn_scale || !a->bn_shift) return TL_ERR_ARG;
      if (a->bn_x_ld % 8 || ((uintptr_t)a->bn_x) % 16 || ((uintptr_t)a->bn_mean) % 16 || ((uintptr_t)a->bn_rstd) % 16 || ((uintptr_t)a->bn_scale) % 16 ||
          ((uintptr_t)a->bn_shift) % 16)
        return TL_ERR_UNSUPPORTED;
    }
    if (a->red_nparts) *a->red_nparts = 0;
  }
  hipStream_t s = tl_s(stream);
  if (p.one_hot == 2) {
    // packed one-hot table (tl_level.inv_packed): only the gather-once form of the direct kernel decodes it (16-bit, or fp32 rows with split-bf16
    // weights; 8 taps, 64 -> 32) -- everything else would read it as an [8][n] table
    const bool shape = a->K == 8 && a->Cin == 64 && a->Cout == 32 && !train && !a->in_scale && !a->in_relu && g_direct && g_direct_oh;
    if (!shape || !(dt == TL_BF16 || (dt == TL_F32 && p.w_x3))) return TL_ERR_UNSUPPORTED;
    return L_direct(p, dt, s);
  }
  const bool vec_ok = (a->in_ld % 8 == 0) && (((uintptr_t)a->in) % 16 == 0) && (((uintptr_t)a->weight) % 16 == 0);
  const bool out_vec = (a->out_ld % 8 == 0) && (((uintptr_t)a->out) % 16 == 0) &&
                       (!a->out2 || (a->out2_ld % 8 == 0 && ((uintptr_t)a->out2) % 16 == 0)) && (!a->out3 || (a->out3_ld % 8 == 0 && ((uintptr_t)a->out3) % 16 == 0));
  if (a->in_all_ones && !train && (dt == TL_BF16 || (dt == TL_F32 && !f16)) && out_vec && (g_direct || !a->table) && ((uintptr_t)a->weight) % 4 == 0 && !a->residual) {
    const int rc = L_ones27(p, dt, s);
    if (rc != TL_ERR_UNSUPPORTED) return rc;
  }
  if (has_blk && dt == TL_BF16 && g_blk && vec_ok && out_vec && (!a->residual || (a->res_ld % 8 == 0 && ((uintptr_t)a->residual) % 16 == 0)) &&
      (!a->out_scale || (((uintptr_t)a->out_scale) % 4 == 0 && ((uintptr_t)a->out_shift) % 4 == 0)) && (!a->in_relu || a->in_scale)) {
    const int rc = L_blk(p, s);
    if (rc != TL_ERR_UNSUPPORTED) return rc;
  }
  if (has_blk && a->dtype == TL_F32 && p.w_x3 && g_blk && !train) {             // bf16x3 on block-local rows: the staged-unit kernel for fp32 rows
    const int rc = tl_launch_conv_blk_x3(p, s);
    if (rc != TL_ERR_UNSUPPORTED) return rc;
  }
  if (!a->table && a->K != 1) return TL_ERR_UNSUPPORTED;       // block-local rows without a shape the staged-unit kernel serves
  if (a->table_scatter && p.one_hot && g_up && !train && dt == TL_BF16 && vec_ok && out_vec &&
      (!a->out_scale || (((uintptr_t)a->out_scale) % 16 == 0 && ((uintptr_t)a->out_shift) % 16 == 0))) {
    const int rc = L_up(p, a->table_scatter, s);          // inverse conv walking the coarse rows (levels 1-4)
    if (rc != TL_ERR_UNSUPPORTED) return rc;
  }
  if (dt == TL_BF16 && a->Cin == 4 && a->Cout == 32 && g_direct && out_vec && ((uintptr_t)a->in) % 8 == 0 && ((uintptr_t)a->weight) % 16 == 0 &&
      (!a->residual || (a->res_ld % 8 == 0 && ((uintptr_t)a->residual) % 16 == 0))) {
    const int rc = L_direct(p, TL_BF16, s);
    if (rc != TL_ERR_UNSUPPORTED) return rc;
  }
  if (!train && a->Cin <= 8 && a->Cout % 8 == 0 && a->Cout <= 64 && !a->in_scale && !a->in_relu && out_vec) return L_tinycin(p, dt, s);
  if (!train && vec_ok && a->n_out <= g_small_rows && a->Cin % 32 == 0 && a->Cout % 32 == 0) return L_small(p, dt, s);
  if (train && a->n_out <= g_small_rows) return TL_ERR_UNSUPPORTED;            // small levels: the separate passes
  const bool aligned = (a->in_ld % 4 == 0) && (((uintptr_t)a->in) % 16 == 0) && (((uintptr_t)a->weight) % 16 == 0) &&
                       (!a->in_scale || (((uintptr_t)a->in_scale) % 16 == 0 && ((uintptr_t)a->in_shift) % 16 == 0));
  if (p.w_x3 && a->Cin >= 256) {
    // split-bf16 weights of a wide conv come as two half-width convs (tl_pack_weight_x3): the decoder's 2C -> C conv of level 4 (256 -> 128)
    // runs as two 128 -> 128 launches, the first leaving its raw sums in `out`, the second taking them as its residual
    if (dt == TL_F32 && !train && g_stream && aligned && out_vec && a->Cin % 64 == 0 && a->Cout % 32 == 0 && !a->in_scale && !a->in_relu && !a->residual &&
        !a->out2 && !a->out3 && a->n_out > g_small_rows && (!a->out_scale || (((uintptr_t)a->out_scale) % 16 == 0 && ((uintptr_t)a->out_shift) % 16 == 0))) {
      ConvP p1 = p;
      p1.Cin = a->Cin / 2; p1.out_scale = nullptr; p1.out_shift = nullptr; p1.out_relu = 0;
      const int rc1 = L_stream(p1, TL_F32, s);
      if (rc1 == TL_OK) {
        ConvP p2 = p;
        p2.Cin = a->Cin / 2; p2.in = static_cast<const float*>(a->in) + a->Cin / 2;
        p2.w_x3 = static_cast<const char*>(p.w_x3) + (size_t)a->K * a->Cout * (a->Cin / 2) * 4;
        p2.res = a->out; p2.res_ld = a->out_ld;
        return L_stream(p2, TL_F32, s);
      }
      if (rc1 != TL_ERR_UNSUPPORTED) return rc1;
    }
    p.w_x3 = nullptr;                                            // (the half layout is not what the single-launch kernels read)
  }
  if (dt == TL_F32 && aligned && out_vec && a->Cin % 32 == 0 && a->Cout % 32 == 0 && !a->in_scale && !a->in_relu &&
      (!a->residual || (a->res_ld % 4 == 0 && ((uintptr_t)a->residual) % 16 == 0)) &&
      (!a->out_scale || (((uintptr_t)a->out_scale) % 16 == 0 && ((uintptr_t)a->out_shift) % 16 == 0))) {
    if (g_direct) { const int rc = L_direct(p, TL_F32, s); if (rc != TL_ERR_UNSUPPORTED) return rc; }
    if (g_stream && g_streamq && g_streamq_x3 && p.w_x3 && !train && a->n_out > g_small_rows) {     // bf16x3: quad-coalesced gathers where the shape has an instantiation
      const int rc = tl_launch_conv_streamq_x3(p, g_streamq_x3, s);
      if (rc != TL_ERR_UNSUPPORTED) return rc;
    }
    if (g_stream) { const int rc = L_stream(p, TL_F32, s); if (rc != TL_ERR_UNSUPPORTED) return rc; }
  }
  if (train && dt == TL_F32) return TL_ERR_UNSUPPORTED;
  // fp32 shapes none of the resident-weight / streamed-weight kernels has an instantiation for (the level-4 <- 5 inverse conv 160 -> 128 on 38 k
  // rows): the small-level kernel before the tile kernel while the level is still small (0.198 -> 0.082 ms in the parity-fast mode)
  if (dt == TL_F32 && !f16 && vec_ok && a->n_out <= 4 * g_small_rows && a->Cin % 32 == 0 && a->Cout % 32 == 0 && !a->in_scale && !a->in_relu)
    return L_small(p, dt, s);
  if (dt == TL_F32 && aligned && a->Cin % KC == 0 && a->Cout % 32 == 0 && a->Cout <= 224) {
    switch (a->Cout / 32) {
      case 1: return launch_mfma_f32<1>(p, s);
      case 2: return launch_mfma_f32<2>(p, s);
      case 3: return launch_mfma_f32<3>(p, s);
      case 4: return launch_mfma_f32<4>(p, s);
      case 5: return launch_mfma_f32<5>(p, s);
      case 6: return launch_mfma_f32<6>(p, s);
      case 7: return launch_mfma_f32<7>(p, s);
    }
  }
  if (dt == TL_BF16 && aligned && vec_ok && out_vec && a->Cin % 32 == 0 && a->Cout % 32 == 0 && a->Cout <= 224 &&
      (!a->residual || (a->res_ld % 8 == 0 && ((uintptr_t)a->residual) % 16 == 0)) &&
      (!a->out_scale || (((uintptr_t)a->out_scale) % 16 == 0 && ((uintptr_t)a->out_shift) % 16 == 0)))
  {
    if (p.one_hot && g_direct && g_direct_oh && a->K == 8 && a->Cin == 64 && a->Cout == 32) {   // level-1 inverse conv: weights resident, no barriers
      const int rc = L_direct(p, TL_BF16, s);
      if (rc != TL_ERR_UNSUPPORTED) return rc;
    }
    if (p.one_hot && g_stream && a->K == 8) {                  // inverse convs: the stream kernel's gather-once form
      const int rc = L_stream(p, TL_BF16, s);
      if (rc != TL_ERR_UNSUPPORTED) return rc;
    }
#ifdef TL_DEV
    if (!train && g_win && a->K == 27 && a->n_out >= g_win_min_rows && ((a->Cin >= 64 && a->Cout >= 64) || g_win >= 2)) {
      const int rc = tl_launch_conv_win(p, s);
      if (rc != TL_ERR_UNSUPPORTED) return rc;
    }
#endif
    if (g_direct) {
      const int rc = L_direct(p, TL_BF16, s);
      if (rc != TL_ERR_UNSUPPORTED) return rc;
    }
    if (g_stream && g_streamq) {
      const int rc = L_streamq(p, s);
      if (rc != TL_ERR_UNSUPPORTED) return rc;
    }
    if (g_stream) {
      const int rc = L_stream(p, TL_BF16, s);
      if (rc != TL_ERR_UNSUPPORTED) return rc;
    }
    if (train) return TL_ERR_UNSUPPORTED;
    return f16 ? tl_launch_conv_bf16_f16(p, g_bf16_depth, g_bf16_units, s) : tl_launch_conv_bf16(p, g_bf16_depth, g_bf16_units, s);
  }
  if (train || f16) return TL_ERR_UNSUPPORTED;             // scalar fallbacks: TL_F32 / TL_BF16 only
  {                                                              // 1x1 with <= 8 output channels (the heads' output Linears in training)
    const int rc = tl_launch_conv_tinycout(p, dt, s);
    if (rc != TL_ERR_UNSUPPORTED) return rc;
  }
  const unsigned g = tl_grid(a->n_out * a->Cout, 256);
  if (dt == TL_F32) k_conv_generic<float><<<g, 256, 0, s>>>(p);
  else k_conv_generic<__hip_bfloat16><<<g, 256, 0, s>>>(p);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int64_t tl_conv_red_parts(int64_t n_out) {
  // the stream kernels write one row per 256 output rows, the persistent direct kernels at most 2048
  const int64_t a = tl_cdiv(n_out > 0 ? n_out : 1, 256);
  return a > 2048 ? a : 2048;
}

int tl_pack_weight(const float* w_ref, int Cout, int K, int Cin, void* w_packed, int dtype, tl_stream_t stream) {
  if (!w_ref || !w_packed || Cout <= 0 || K <= 0 || Cin <= 0 || (dtype != TL_F32 && dtype != TL_BF16 && dtype != TL_F16)) return TL_ERR_ARG;
  k_pack_weight<<<tl_grid((int64_t)Cout * K * Cin, 256), 256, 0, tl_s(stream)>>>(w_ref, Cout, K, Cin, w_packed, dtype);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int64_t tl_pack_weight_x3_bytes(int Cout, int K, int Cin) {
  if (Cout <= 0 || K <= 0 || Cin <= 0 || Cin % 32) return -1;
  const int64_t one = (int64_t)K * Cout * Cin * 4;
  return (Cin < 256 && Cout % 32 == 0) ? 2 * one : one;
}

int tl_pack_weight_x3(const float* w_ref, int Cout, int K, int Cin, void* w_x3, tl_stream_t stream) {
  if (!w_ref || !w_x3 || Cout <= 0 || K <= 0 || Cin <= 0 || Cin % 32) return TL_ERR_ARG;
  k_pack_weight_x3<<<tl_grid((int64_t)Cout * K * Cin * 2, 256), 256, 0, tl_s(stream)>>>(w_ref, Cout, K, Cin, reinterpret_cast<uint16_t*>(w_x3));
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_pack_weight_dgrad(const float* w_ref, int Cout, int K, int Cin, int flip, void* w_t, int dtype, tl_stream_t stream) {
  if (!w_ref || !w_t || Cout <= 0 || K <= 0 || Cin <= 0 || (dtype != TL_F32 && dtype != TL_BF16 && dtype != TL_F16)) return TL_ERR_ARG;
  k_pack_weight_dgrad<<<tl_grid((int64_t)Cout * K * Cin, 256), 256, 0, tl_s(stream)>>>(w_ref, Cout, K, Cin, flip, w_t, dtype);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_pack_weights_batch(const tl_pack_desc* descs, const int32_t* blocks, int64_t n_blocks, int dtype, tl_stream_t stream) {
  if (!descs || !blocks || n_blocks <= 0 || n_blocks > 0x7FFFFFFF || (dtype != TL_F32 && dtype != TL_BF16 && dtype != TL_F16)) return TL_ERR_ARG;
  k_pack_batch<<<(unsigned)n_blocks, 256, 0, tl_s(stream)>>>(descs, reinterpret_cast<const int2*>(blocks), dtype);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_pack_weight_frag(const float* w_ref, int Cout, int K, int Cin, void* w_frag, int dtype, tl_stream_t stream) {
  if (!w_ref || !w_frag || Cout <= 0 || K <= 0 || Cin <= 0 || Cout % 32 || Cin % 32 || (dtype != TL_F32 && dtype != TL_BF16 && dtype != TL_F16)) return TL_ERR_ARG;
  k_pack_weight_frag<<<tl_grid((int64_t)Cout * K * Cin / 8, 256), 256, 0, tl_s(stream)>>>(w_ref, Cout, K, Cin, w_frag, dtype);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_affine_relu(const void* in, int64_t in_ld, void* out, int64_t out_ld, int64_t n, int C, int dtype, const float* scale,
                   const float* shift, int relu, tl_stream_t stream) {
  if (!in || !out || n <= 0 || C <= 0 || (scale == nullptr) != (shift == nullptr)) return TL_ERR_ARG;
  if ((dtype == TL_F32 || dtype == TL_BF16 || dtype == TL_F16) && C % 8 == 0 && in_ld % 8 == 0 && out_ld % 8 == 0 && ((uintptr_t)in) % 16 == 0 && ((uintptr_t)out) % 16 == 0) {
    const int vpr = C / 8;
    int64_t gv = tl_cdiv(n * vpr, 256 * 4);                      // about four vectors per thread, at most 16 workgroups per CU
    if (gv > 256 * 16) gv = 256 * 16;
    gv = tl_cdiv(gv * 256, (int64_t)vpr * 256) * vpr;           // grid * 256 must be a multiple of the vectors per row
    if (dtype == TL_BF16) k_affine_relu_v8<1><<<(unsigned)gv, 256, 0, tl_s(stream)>>>(in, in_ld, out, out_ld, n, C, scale, shift, relu);
    else if (dtype == TL_F16) k_affine_relu_v8<2><<<(unsigned)gv, 256, 0, tl_s(stream)>>>(in, in_ld, out, out_ld, n, C, scale, shift, relu);
    else k_affine_relu_v8<0><<<(unsigned)gv, 256, 0, tl_s(stream)>>>(in, in_ld, out, out_ld, n, C, scale, shift, relu);
    TL_CHECK_LAUNCH();
    return TL_OK;
  }
  const unsigned g = tl_grid(n * C, 256);
  if (dtype == TL_F32) k_affine_relu<float><<<g, 256, 0, tl_s(stream)>>>((const float*)in, in_ld, (float*)out, out_ld, n, C, scale, shift, relu);
  else if (dtype == TL_BF16) k_affine_relu<__hip_bfloat16><<<g, 256, 0, tl_s(stream)>>>((const __hip_bfloat16*)in, in_ld, (__hip_bfloat16*)out, out_ld, n, C, scale, shift, relu);
  else if (dtype == TL_F16) k_affine_relu<_Float16><<<g, 256, 0, tl_s(stream)>>>((const _Float16*)in, in_ld, (_Float16*)out, out_ld, n, C, scale, shift, relu);
  else return TL_ERR_ARG;
  TL_CHECK_LAUNCH();
  return TL_OK;
}

}  // extern "C"
