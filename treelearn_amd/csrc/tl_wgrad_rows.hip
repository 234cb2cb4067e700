// Weight gradient of the 1x1 "convs" -- the i_branch of the decoder's first tail block (blocks.py:29-39 Custom1x1Subm3d: 2C -> C at every
// level) and the hidden Linears of the two heads (blocks.py:8-26, 32 -> 32 over all points) -- bf16 operands, fp32 accumulation:
//     gW[co][ci] = sum_o gout[o][co] * x[o][ci]
// A plain [Cout x rows] x [rows x Cin] GEMM whose contraction index is the row: HBM-bound (both operands are read once, 0.1-0.2 ms at
// level 1), but the pair-list kernel of tl_wgrad.hip walked it through its rulebook machinery (ballots, compaction lists, 16-pair
// batches).  Here every wave streams its own 16 / 32-row steps: full-row 16-B loads of gout and x (PD steps in flight), two wave-private
// XOR-swizzled LDS tiles, ds_read_b64_tr_b16 for both operands (the layout rules of tl_wgrad_dense.hip), NBO x NBIW MFMAs per 16 rows.
// Cin is cut into slices of NBIW * 32 channels over blockIdx.y; the waves of a workgroup add their tiles in wave order through LDS, the
// workgroups' partials go through the ordered reduction (tl_launch_wgrad_reduce).  Deterministic.
#include "tl_conv_internal.h"
#include "tl_f16_train.h"
#include <atomic>

int tl_launch_wgrad_reduce(const float* ws, int64_t nparts, int64_t per, float* gw, hipStream_t s, int K = 1, int Cout = 0, int Cin = 0, int ref_layout = 0);   // tl_wgrad_dense.hip

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4* lds4;

template <int P>
static __device__ __forceinline__ int swz(int r) {              // as in tl_wgrad_dense.hip
  if constexpr ((P / 64) % 2 == 1) return 0;
  else if constexpr ((P / 64) % 4 == 2) return ((r >> 1) & 1) << 2;
  else return (r & 3) << 2;
}

constexpr int kRW = 8;                                          // waves per workgroup

template <int NBO, int NBIW, int KS, int PD>
__global__ void __launch_bounds__(kRW * 64) k_wgrad_rows(const uint16_t* __restrict__ x, int64_t x_ld, const uint16_t* __restrict__ g, int64_t g_ld, int64_t n,
                                                        int Cin, float* __restrict__ ws) {
  constexpr int COUT = NBO * 32, CS = NBIW * 32, PG = COUT * 2, PX = CS * 2, ROWS = 16 * KS;
  constexpr int PRG = COUT / 8, PRX = CS / 8;
  constexpr int NLG = (ROWS * PRG + 63) / 64, NLX = (ROWS * PRX + 63) / 64;
  constexpr int TILE = ROWS * (PG + PX);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* Gw = smem + wv * (2 * TILE);                            // wave-private [2][ROWS][PG] then [2][ROWS][PX]
  char* Xw = Gw + 2 * ROWS * PG;
  const int slice = blockIdx.y;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, (int)min((int64_t)0x7FFFFFFF, n * x_ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g), 0, (int)min((int64_t)0x7FFFFFFF, n * g_ld * 2), 0x00020000);
  const unsigned x_ld_b = (unsigned)(x_ld * 2), g_ld_b = (unsigned)(g_ld * 2);
  const int fi = lane & 31, fh = lane >> 5, ti = lane & 15, tg = (lane >> 4) & 1;
  const int prow = 8 * fh + (ti >> 2);
  int ga[NBO], xa[NBIW];
#pragma unroll
  for (int a = 0; a < NBO; ++a) ga[a] = prow * PG + (((4 * a + 2 * tg + ((ti & 3) >> 1)) ^ swz<PG>(prow)) << 4) + 8 * (ti & 1);
#pragma unroll
  for (int b = 0; b < NBIW; ++b) xa[b] = prow * PX + (((4 * b + 2 * tg + ((ti & 3) >> 1)) ^ swz<PX>(prow)) << 4) + 8 * (ti & 1);

  f32x16 acc[NBO][NBIW];
#pragma unroll
  for (int a = 0; a < NBO; ++a)
#pragma unroll
    for (int b = 0; b < NBIW; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const int64_t nsteps = (n + ROWS - 1) / ROWS;
  const int64_t first = (int64_t)blockIdx.x * kRW + wv, stride = (int64_t)gridDim.x * kRW;
  auto issue = [&](int64_t st, u32x4 (&gq)[NLG], u32x4 (&xq)[NLX]) __attribute__((always_inline)) {
    const int64_t r0 = st * ROWS;
#pragma unroll
    for (int i = 0; i < NLG; ++i) {
      const int q = lane + 64 * i, r = q / PRG, pc = q % PRG;
      const bool ok = st < nsteps && r0 + r < n && (ROWS * PRG % 64 == 0 || q < ROWS * PRG);
      const unsigned off = ok ? (unsigned)(r0 + r) * g_ld_b + (unsigned)(pc * 16) : 0xFFFFFFFFu;
      gq[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int q = lane + 64 * i, r = q / PRX, pc = q % PRX;
      const bool ok = st < nsteps && r0 + r < n && (ROWS * PRX % 64 == 0 || q < ROWS * PRX);
      const unsigned off = ok ? (unsigned)(r0 + r) * x_ld_b + (unsigned)((slice * CS + pc * 8) * 2) : 0xFFFFFFFFu;
      xq[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0));
    }
  };
  u32x4 gb[PD][NLG], xb[PD][NLX];
#pragma unroll
  for (int d = 0; d < PD; ++d) issue(first + d * stride, gb[d], xb[d]);
  for (int64_t s0 = first; s0 < nsteps; s0 += PD * stride) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      char* Gc = Gw + (d & 1) * (ROWS * PG);
      char* Xc = Xw + (d & 1) * (ROWS * PX);
#pragma unroll
      for (int i = 0; i < NLG; ++i) {
        const int q = lane + 64 * i, r = q / PRG, pc = q % PRG;
        if (ROWS * PRG % 64 == 0 || q < ROWS * PRG) *reinterpret_cast<u32x4*>(Gc + r * PG + ((pc ^ swz<PG>(r)) << 4)) = gb[d][i];
      }
#pragma unroll
      for (int i = 0; i < NLX; ++i) {
        const int q = lane + 64 * i, r = q / PRX, pc = q % PRX;
        if (ROWS * PRX % 64 == 0 || q < ROWS * PRX) *reinterpret_cast<u32x4*>(Xc + r * PX + ((pc ^ swz<PX>(r)) << 4)) = xb[d][i];
      }
      issue(s0 + (d + PD) * stride, gb[d], xb[d]);                  // past the end: every offset out of range, zeros back
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u32x4 A[NBO], B[NBIW];
#pragma unroll
        for (int a = 0; a < NBO; ++a)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(Gc + (16 * ks + 4 * q) * PG + ga[a])));
            A[a][2 * q] = v[0]; A[a][2 * q + 1] = v[1];
          }
#pragma unroll
        for (int b = 0; b < NBIW; ++b)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(Xc + (16 * ks + 4 * q) * PX + xa[b])));
            B[b][2 * q] = v[0]; B[b][2 * q + 1] = v[1];
          }
#pragma unroll
        for (int a = 0; a < NBO; ++a)
#pragma unroll
          for (int b = 0; b < NBIW; ++b)
            acc[a][b] = h16_mfma(A[a], B[b], acc[a][b]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }

  // the waves' tiles are added in wave order through LDS -> one partial per workgroup: ws[blockIdx.x][Cout][Cin]
  __syncthreads();
  float (*Rs)[CS + 1] = reinterpret_cast<float (*)[CS + 1]>(smem);
  static_assert(COUT * (CS + 1) * 4 <= kRW * 2 * TILE, "the reduction tile fits in the staging tiles");
  float* wp = ws + ((int64_t)blockIdx.x * COUT) * Cin + slice * CS;
  for (int w = 0; w < kRW; ++w) {
    if (wv == w) {
#pragma unroll
      for (int a = 0; a < NBO; ++a)
#pragma unroll
        for (int b = 0; b < NBIW; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int col = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh, cil = b * 32 + fi;
            float v = acc[a][b][r];
            if (w > 0) v += Rs[col][cil];
            if (w < kRW - 1) Rs[col][cil] = v;
            else wp[(int64_t)col * Cin + cil] = v;
          }
    }
    __syncthreads();
  }
}

// The 4-channel input conv (tree_learn.py:37-39 `input_conv`, K = 27, Cin = 4, Cout = 32): gW[k][co][ci] = sum_o gout[o][co] * x[nbr[k][o]][ci].
// With (tap, channel) = 108 (padded to 128) as ONE "input channel" axis this is the same row-streaming GEMM: per 16-row step a wave
// gathers the 27 neighbours' 8-byte rows (absent = index -1 = out of range = zeros) into a [16][256 B] tile next to the gout tile and
// runs four MFMAs.  The fp32-MFMA pair kernel took 1.3 ms for this layer; this form is bound by the 108 B / voxel rulebook read.
__global__ void __launch_bounds__(kRW * 64) k_wgrad_in4(const uint16_t* __restrict__ x, int64_t x_ld, const uint16_t* __restrict__ g, int64_t g_ld,
                                                       const int32_t* __restrict__ table, int64_t n, int64_t n_in, float* __restrict__ ws) {
  constexpr int K = 27, PG = 64, PX = 256, ROWS = 16, NI = (ROWS * K + 63) / 64, PD = 2;
  constexpr int TILE = ROWS * (PG + PX);
  __shared__ __attribute__((aligned(16))) char smem[kRW * 2 * TILE];
  typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* Gw = smem + wv * (2 * TILE);
  char* Xw = Gw + 2 * ROWS * PG;
  for (int e = lane; e < 2 * ROWS * PX / 16; e += 64) *reinterpret_cast<u32x4*>(Xw + e * 16) = u32x4{0u, 0u, 0u, 0u};     // the 20 padding channels stay zero
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, (int)min((int64_t)0x7FFFFFFF, (n_in - 1) * x_ld * 2 + 8), 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g), 0, (int)min((int64_t)0x7FFFFFFF, n * g_ld * 2), 0x00020000);
  const unsigned x_ld_b = (unsigned)(x_ld * 2), g_ld_b = (unsigned)(g_ld * 2);
  const int fi = lane & 31, fh = lane >> 5, ti = lane & 15, tg = (lane >> 4) & 1;
  const int prow = 8 * fh + (ti >> 2);
  const int ga = prow * PG + ((2 * tg + ((ti & 3) >> 1)) << 4) + 8 * (ti & 1);
  int xa[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) xa[b] = prow * PX + (((4 * b + 2 * tg + ((ti & 3) >> 1)) ^ swz<PX>(prow)) << 4) + 8 * (ti & 1);
  int xr[NI], xk[NI], xl[NI];                                   // this lane's (row, tap) items of a step and their place in the tile
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = lane + 64 * i;
    xr[i] = q / K; xk[i] = q % K;
    xl[i] = xr[i] * PX + ((((xk[i] >> 1)) ^ swz<PX>(xr[i])) << 4) + 8 * (xk[i] & 1);
  }
  f32x16 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
  const int64_t nsteps = (n + ROWS - 1) / ROWS;
  const int64_t first = (int64_t)blockIdx.x * kRW + wv, stride = (int64_t)gridDim.x * kRW;
  auto idx_load = [&](int64_t st, int (&dst)[NI]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int64_t row = st * ROWS + xr[i];
      dst[i] = (st < nsteps && row < n && lane + 64 * i < ROWS * K) ? table[(int64_t)xk[i] * n + row] : -1;
    }
  };
  auto issue = [&](int64_t st, const int (&idx)[NI], u32x4& gq, u32x2_ (&xq)[NI]) __attribute__((always_inline)) {
    const int64_t row = st * ROWS + (lane >> 2);
    const unsigned off = (st < nsteps && row < n) ? (unsigned)row * g_ld_b + (unsigned)((lane & 3) * 16) : 0xFFFFFFFFu;
    gq = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)off, 0, 0));
#pragma unroll
    for (int i = 0; i < NI; ++i) xq[i] = __builtin_bit_cast(u32x2_, __builtin_amdgcn_raw_buffer_load_b64(rx, (int)((unsigned)idx[i] * x_ld_b), 0, 0));
  };
  int ia[PD][NI];
  u32x4 gb[PD]; u32x2_ xb[PD][NI];
#pragma unroll
  for (int d = 0; d < PD; ++d) idx_load(first + d * stride, ia[d]);
#pragma unroll
  for (int d = 0; d < PD; ++d) { issue(first + d * stride, ia[d], gb[d], xb[d]); idx_load(first + (d + PD) * stride, ia[d]); }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int64_t s0 = first; s0 < nsteps; s0 += PD * stride) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      char* Gc = Gw + (d & 1) * (ROWS * PG);
      char* Xc = Xw + (d & 1) * (ROWS * PX);
      *reinterpret_cast<u32x4*>(Gc + (lane >> 2) * PG + ((lane & 3) << 4)) = gb[d];
#pragma unroll
      for (int i = 0; i < NI; ++i)
        if (ROWS * K % 64 == 0 || lane + 64 * i < ROWS * K) *reinterpret_cast<u32x2_*>(Xc + xl[i]) = xb[d][i];
      issue(s0 + (d + PD) * stride, ia[d], gb[d], xb[d]);           // entries requested one round earlier
      idx_load(s0 + (d + 2 * PD) * stride, ia[d]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      u32x4 A, B[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(Gc + 4 * q * PG + ga)));
        A[2 * q] = v[0]; A[2 * q + 1] = v[1];
      }
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(Xc + 4 * q * PX + xa[b])));
          B[b][2 * q] = v[0]; B[b][2 * q + 1] = v[1];
        }
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = h16_mfma(A, B[b], acc[b]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  // waves added in wave order -> ws[blockIdx.x][k][co][ci], (k, ci) = the MFMA column n = 4 k + ci
  __syncthreads();
  float (*Rs)[129] = reinterpret_cast<float (*)[129]>(smem);
  float* wp = ws + (int64_t)blockIdx.x * (K * 32 * 4);
  for (int w = 0; w < kRW; ++w) {
    if (wv == w) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (r & 3) + 8 * (r >> 2) + 4 * fh, nn = b * 32 + fi;
          float v = acc[b][r];
          if (w > 0) v += Rs[co][nn];
          if (w < kRW - 1) Rs[co][nn] = v;
          else if (nn < 4 * K) wp[((nn >> 2) * 32 + co) * 4 + (nn & 3)] = v;
        }
    }
    __syncthreads();
  }
}

template <int NBO, int NBIW, int KS, int PD>
int launch(const uint16_t* x, int64_t x_ld, const uint16_t* g, int64_t g_ld, int64_t n, int Cin, int gx, float* ws, hipStream_t s) {
  const size_t lds = (size_t)kRW * 2 * (16 * KS) * (NBO * 64 + NBIW * 64);
  if (lds > 160 * 1024) return TL_ERR_UNSUPPORTED;
  static TlAttrOnce attr_once;                     // per kernel instantiation AND device (the attribute is per device)
  if (!tl_lds_attr(attr_once, reinterpret_cast<const void*>(&k_wgrad_rows<NBO, NBIW, KS, PD>), 160 * 1024)) return TL_ERR_LAUNCH;
  k_wgrad_rows<NBO, NBIW, KS, PD><<<dim3((unsigned)gx, (unsigned)(Cin / (NBIW * 32))), kRW * 64, lds, s>>>(x, x_ld, g, g_ld, n, Cin, ws);
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

}  // namespace

int g_wgrad_rows = 1;             // tl_set_tuning("wgrad_rows", 0): the pair-list kernels for K = 1 as well

// partial tile sets (workgroups along x) of the row-streaming form for a K = 1 shape, 0 if the shape is not served
int tl_wgrad_rows_parts(int64_t n, int Cin, int Cout) {
  if (!g_wgrad_rows || n < 30000 || Cout % 32 || Cin % 32 || Cout > 128 || Cin > 256) return 0;
  const int ns = Cin / (Cout >= 96 ? 32 : (Cin % 64 == 0 ? 64 : 32));
  int gx = 512 / ns;                                            // about two workgroups per CU
  if (gx < 16) gx = 16;
  const int64_t cap = (n / 32 + 7) / 8;                         // every wave gets at least one step
  return (int)(gx < cap ? gx : cap);
}

int tl_launch_wgrad_rows(const uint16_t* x, int64_t x_ld, const uint16_t* g, int64_t g_ld, int64_t n, int Cin, int Cout, float* gw, float* ws, hipStream_t s) {
  const int gx = tl_wgrad_rows_parts(n, Cin, Cout);
  if (!gx) return TL_ERR_UNSUPPORTED;
  int rc = TL_ERR_UNSUPPORTED;
  const bool w2 = Cin % 64 == 0;
  switch (Cout / 32) {
    case 1: rc = w2 ? launch<1, 2, 2, 2>(x, x_ld, g, g_ld, n, Cin, gx, ws, s) : launch<1, 1, 2, 2>(x, x_ld, g, g_ld, n, Cin, gx, ws, s); break;
    case 2: rc = w2 ? launch<2, 2, 2, 2>(x, x_ld, g, g_ld, n, Cin, gx, ws, s) : launch<2, 1, 2, 2>(x, x_ld, g, g_ld, n, Cin, gx, ws, s); break;
    case 3: rc = launch<3, 1, 1, 2>(x, x_ld, g, g_ld, n, Cin, gx, ws, s); break;
    case 4: rc = launch<4, 1, 1, 2>(x, x_ld, g, g_ld, n, Cin, gx, ws, s); break;
  }
  if (rc != TL_OK) return rc;
  return tl_launch_wgrad_reduce(ws, gx, (int64_t)Cout * Cin, gw, s);
}

// the 4 -> 32 input conv (K = 27): partial tile sets / launch
int tl_wgrad_in4_parts(int64_t n, int K, int Cin, int Cout) {
  if (!g_wgrad_rows || K != 27 || Cin != 4 || Cout != 32 || n < 30000) return 0;
  const int64_t cap = (n / 16 + 7) / 8;
  return (int)(512 < cap ? 512 : cap);
}

int tl_launch_wgrad_in4(const uint16_t* x, int64_t x_ld, const uint16_t* g, int64_t g_ld, const int32_t* table, int64_t n, int64_t n_in, float* gw, float* ws,
                        hipStream_t s, int ref_layout) {
  const int gx = tl_wgrad_in4_parts(n, 27, 4, 32);
  if (!gx || !table || x_ld % 4 || ((uintptr_t)x) % 8 || g_ld % 8 || ((uintptr_t)g) % 16 || (n_in - 1) * x_ld * 2 + 8 > 0x7FFF0000ll) return TL_ERR_UNSUPPORTED;
  k_wgrad_in4<<<gx, kRW * 64, 0, s>>>(x, x_ld, g, g_ld, table, n, n_in, ws);
  if (hipGetLastError() != hipSuccess) return TL_ERR_LAUNCH;
  return tl_launch_wgrad_reduce(ws, gx, 27 * 32 * 4, gw, s, 27, 32, 4, ref_layout);
}
