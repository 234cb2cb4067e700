// Inverse (transposed, stride-2) sparse conv, COARSE-stationary: reference blocks.py:113-123 `SparseInverseConv3d(kernel_size=2)` -- every fine
// row has exactly one parent row and one tap, out[child[k][c]] = W[k] . x[c].
//
// The gather forms walk the FINE rows (the table's one valid entry per row): each of them fetches its parent's row (8 x the coarse tensor
// through L2, 64..192-B gathers) and all K taps are contracted for every fine row, 7 of 8 against zeros.  Here a wave owns 32 consecutive
// COARSE rows: their features are read once, coalesced, straight into A fragments; for each tap the 32 x Cout product is formed and its rows
// are scattered to the children that exist (tl_conv_args.table_scatter = the stride-2 conv's own rulebook, i32[K][n_in]) -- each fine row is
// written exactly once, whole rows per view.  MFMA work: n_coarse x K row-taps instead of n_fine x K (4.3 x less at level 2 of config 2),
// no gathers.  Weights: the fragment-order copy (tl_pack_weight_frag), resident in LDS.  Results are bit-identical to the gather forms: one
// product per output element, same k order.
//
// Measured on the config-2 tile (inside the forward, us): 96 -> 64 (level 2 <- 3, two 128-B views) 138 -> 102; 64 -> 32 (level 1 <- 2, one
// 64-B view) 78 -> 117 -- scattered 64-B row writes cost more than the gather form's parent fetches, which hit L2; with the weights read
// from L2 per tile (they do not fit the LDS) 128 -> 96 58 -> 154 and 160 -> 128 46 -> 214.  So: one shape, the one with wide rows and two
// views; the others stay on the gather forms.
#include "tl_conv_internal.h"
#include <atomic>

namespace {

template <int NBI, int NBO, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_conv_up(ConvP p, const int32_t* __restrict__ child, int ntiles) {
  constexpr int K = 8, KS = 2 * NBI;                 // 16-wide contraction steps per tap
  constexpr int EP = 36;                             // epilogue tile [32][36] fp32 per wave: one 32-column block at a time
  constexpr int WB = K * NBO * NBI * 2 * 1024;      // the whole weight tensor, fragment order
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.w_frag);
    u32x4* dst = reinterpret_cast<u32x4*>(smem);
    for (int e = tid; e < WB / 16; e += WAVES * 64) dst[e] = src[e];
    __syncthreads();
  }
  float* ew = reinterpret_cast<float*>(smem + WB) + wv * 32 * EP;
  const unsigned in_ldb = (unsigned)(p.in_ld * 2);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)((p.n_in - 1) * (int64_t)in_ldb + NBI * 64), 0x00020000);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(child), 0, (int)((int64_t)K * p.n_in * 4), 0x00020000);

  const int nw = (int)gridDim.x * WAVES;
  int t = (int)blockIdx.x * WAVES + wv;
  u32x4 A[KS]; int ch[K][2];
  // a tile's loads: the lane's row (fi) of the 32 coarse rows, 16 B per step; the children of the rows this lane stores in the epilogue
  // (rows (lane >> 2) and (lane >> 2) + 16); rows past the end read zeros / are masked by the bounds of the child table
  auto request = [&](int tt, u32x4 (&a)[KS], int (&c)[K][2]) __attribute__((always_inline)) {
    const unsigned row = (unsigned)tt * 32u + (unsigned)fi;
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_)
      a[s_] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(row * in_ldb + (unsigned)(s_ * 32 + fh * 16)), 0, 0));
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int e0 = 0; e0 < 2; ++e0) {
        const int64_t r = (int64_t)tt * 32 + (lane >> 2) + 16 * e0;
        c[k][e0] = r < p.n_in ? __builtin_amdgcn_raw_buffer_load_b32(rc, (int)(((int64_t)k * p.n_in + r) * 4), 0, 0) : -1;
      }
  };
  if (t < ntiles) request(t, A, ch);
  for (; t < ntiles; t += nw) {
    u32x4 An[KS]; int chn[K][2];
    const int tn = t + nw;
    if (tn < ntiles) request(tn, An, chn);           // the next tile's loads are in flight during this tile's 8 taps
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const bool any = __builtin_amdgcn_ballot_w64(ch[k][0] >= 0 || ch[k][1] >= 0) != 0ull;
      if (!any) continue;                            // no coarse row of the tile has this child
#pragma unroll
      for (int nb = 0; nb < NBO; ++nb) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
          // fragment order: vector ((((k * NBO + nb) * NBI + ch) * 2 + j) * 64 + lane), ch = s_ / 2, j = s_ & 1
          const int v = (((k * NBO + nb) * NBI + (s_ >> 1)) * 2 + (s_ & 1)) * 64 + lane;
          acc = h16_mfma(A[s_], *reinterpret_cast<const u32x4*>(smem + (size_t)v * 16), acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) ew[((r & 3) + 8 * (r >> 2) + 4 * fh) * EP + fi] = acc[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e0 = 0; e0 < 2; ++e0) {
          const int rr = (lane >> 2) + 16 * e0, cvv = lane & 3;
          const int orow = ch[k][e0];
          if (orow >= 0 && orow < (int)p.n_out) {           // (a child index outside the output is ignored, never written)
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8), v1 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            epi_views8<true>(p, (int64_t)orow, nb * 32 + cvv * 8, v);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (tn < ntiles) {
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_) A[s_] = An[s_];
#pragma unroll
      for (int k = 0; k < K; ++k) { ch[k][0] = chn[k][0]; ch[k][1] = chn[k][1]; }
    }
  }
}

template <int NBI, int NBO, int WAVES>
int launch_up(const ConvP& p, const int32_t* child, hipStream_t s) {
  const size_t lds = (size_t)8 * NBO * NBI * 2 * 1024 + (size_t)WAVES * 32 * 36 * 4;
  static_assert(8 * NBO * NBI * 2 * 1024 + WAVES * 32 * 36 * 4 <= 160 * 1024, "weights + epilogue tiles must fit the LDS");
  static TlAttrOnce attr_once;                     // per kernel instantiation AND device (the attribute is per device)
  if (!tl_lds_attr(attr_once, reinterpret_cast<const void*>(&k_conv_up<NBI, NBO, WAVES>), 160 * 1024)) return TL_ERR_LAUNCH;
  const int ntiles = (int)tl_cdiv(p.n_in, 32);
  const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  int grid = 256 * (per_cu > 4 ? 4 : per_cu);
  const int need = (int)tl_cdiv(ntiles, WAVES);
  if (grid > need) grid = need;
  k_conv_up<NBI, NBO, WAVES><<<grid, WAVES * 64, lds, s>>>(p, child, ntiles);
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

}  // namespace

// Eligibility beyond tl_conv_fwd's alignment rules: 16-bit, K = 8, the one-hot table's scatter form given, fragment-order weights, no
// gather-side prologue, no residual, inference epilogues; 32-bit byte offsets.
int tl_launch_conv_up(const ConvP& p, const int32_t* child, hipStream_t s) {
  if (!child || !p.w_frag || ((uintptr_t)p.w_frag) % 16 || p.K != 8 || p.in_scale || p.in_relu || p.res || p.epi_mode != TL_EPI_NONE) return TL_ERR_UNSUPPORTED;
  auto big = [&](int64_t rows, int64_t ld) { return (rows - 1) * ld * 2 + 512 >= 0x7FFFFFFFll; };
  if (big(p.n_in, p.in_ld) || big(p.n_out, p.out_ld) || (p.out2 && big(p.n_out, p.out2_ld)) || (p.out3 && big(p.n_out, p.out3_ld)) || (int64_t)8 * p.n_in * 4 >= 0x7FFFFFFFll)
    return TL_ERR_UNSUPPORTED;
  // 96 KB of weights resident; twelve waves (three per SIMD) beside them: 0.055-0.059 / 0.104 ms (one / two views) with 8 waves, 0.050 / 0.100 with
  // 12; 10 and 13 waves -- an uneven load on the four SIMDs -- are slower than 8 (0.060 / 0.121, 0.066 / 0.127)
  if (p.Cin == 96 && p.Cout == 64) return launch_up<3, 2, 12>(p, child, s);
  return TL_ERR_UNSUPPORTED;
}
