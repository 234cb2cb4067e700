// Sparse conv variants for the two shapes the 128-row output-stationary tiling serves badly.
//
// k_conv_small  -- deep U-Net levels (a few hundred .. 16 k voxels, up to 224 channels): a 128-row tile
//   leaves most of the 256 CUs idle and serialises 27 taps x Cin/32 chunks in one workgroup.  Here a
//   workgroup owns 32 rows x (<=64) output channels and its 4 waves split the (tap, chunk) steps between
//   them; MFMA operands are loaded straight from global/L2 into fragment registers (the whole level
//   lives in L2), there is no LDS staging and no barrier in the main loop; the four partial
//   accumulators are reduced through LDS once at the end.  Deterministic (fixed reduction order).
//
// k_conv_tinycin -- the 4-channel input conv (reference tree_learn.py:37-39): HBM-bound on reading the
//   27-tap table and writing [N,32]; weights sit in LDS, one thread = one row x 8 output channels.
#include "tl_conv_internal.h"

namespace {

// FR: B operands come from the fragment-order weight copy (p.w_frag): 1 KB contiguous per load instead of 32 rows x 32 B
// X3 (fp32 rows): split-bf16 contraction on weights in the tl_pack_weight_x3 form (p.w_x3; tl_conv_internal.h: mma16_x3) -- the B "pieces"
// b[..][J] / b[..][2 + J] are then the hi / lo parts of channel group J of the unit
template <bool BF16, int NBB, int WV, bool FR, bool X3 = false>
__global__ void __launch_bounds__(WV * 64) k_conv_small(ConvP p, int ncolblk) {
  static_assert(!X3 || !BF16, "split-bf16: fp32 rows, weights in their own layout (FR: its fragment-order second copy)");
  constexpr int EB = BF16 ? 2 : 4, UB = 32 * EB, NJ = UB / 32;
  constexpr int PF = 4;                                       // steps per batch and wave (their independent loads are issued together)
  constexpr int KMAX = 27;
  __shared__ float red[WV][16][64];                           // partial accumulators of ONE column block at a time
  __shared__ int Is[KMAX * 32];                               // the block's slice of the rulebook, [k][row]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rt = blockIdx.x / ncolblk, cb = blockIdx.x % ncolblk;
  const int fi = lane & 31, fh = lane >> 5;
  const int col0 = cb * NBB * 32;
  const int nchunk = p.Cin / 32;
  const int nsteps = p.K * nchunk;
  const bool pro = p.in_scale != nullptr || p.in_relu;

  // Every (tap, chunk) step starts with a rulebook entry: fetched per step, that is a second dependent memory round trip in
  // front of each batch of gathers.  One coalesced read of the whole [K][32] slice into LDS removes it from the loop.
  for (int e = tid; e < p.K * 32; e += WV * 64) {
    const int64_t r = (int64_t)rt * 32 + (e & 31);
    Is[e] = r < p.n_out ? (p.table ? p.table[(int64_t)(e >> 5) * p.n_out + r] : (int)r) : -1;
  }

  const int in_ld_b = (int)(p.in_ld * EB);
  const int64_t in_bytes = ((int64_t)p.n_in - 1) * in_ld_b + (int64_t)p.Cin * EB;
  const bool buf_ok = in_bytes > 0 && in_bytes + 2 * (int64_t)in_ld_b < 0xFFFFFFFFll;       // else: clamp + mask path
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, buf_ok ? (int)in_bytes : 0, 0x00020000);
  const char* inb = (const char*)p.in;
  const char* Wb = (const char*)(X3 ? p.w_x3 : (FR ? p.w_frag : p.w));
  if constexpr (X3 && FR) Wb += (int64_t)p.K * p.Cout * p.Cin * 4;             // tl_pack_weight_x3: the fragment-order copy follows the slot-order one
  const int CBt = p.Cout / 32;

  f32x16 acc[NBB];
#pragma unroll
  for (int nb = 0; nb < NBB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
  __syncthreads();

  // The wave's steps are s = wv, wv+WV, ...; PF of them form a batch whose loads are all issued before the first MFMA (deep
  // levels are latency-bound: a few hundred rows, 27 x Cin/32 dependent load->MFMA chains per wave).  Requesting the next
  // batch before contracting the current one was measured: l=6,7 -5..-12 %, l=5 +6 % (twice the registers) -- not kept.
  struct Batch { u32x4 a[PF][NJ], b[PF][NBB][NJ]; int idx[PF], ch[PF]; };
  auto request = [&](int s0, Batch& B) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int s = s0 + WV * u;
      const bool sv = s < nsteps;
      const int kk = sv ? s / nchunk : 0;
      B.ch[u] = sv ? s % nchunk : 0;
      B.idx[u] = sv ? Is[kk * 32 + fi] : -1;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const unsigned off = (unsigned)(B.ch[u] * UB + j * 32 + fh * 16);
        if (buf_ok) B.a[u][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((unsigned)B.idx[u] * (unsigned)in_ld_b + off), 0, 0));
        else {
          const u32x4 v = *reinterpret_cast<const u32x4*>(inb + (int64_t)max(B.idx[u], 0) * in_ld_b + off);
          B.a[u][j] = B.idx[u] >= 0 ? v : u32x4{0u, 0u, 0u, 0u};
        }
      }
#pragma unroll
      for (int nb = 0; nb < NBB; ++nb)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if constexpr (FR && X3) B.b[u][nb][j] = *reinterpret_cast<const u32x4*>(Wb + ((((int64_t)(kk * CBt + col0 / 32 + nb) * nchunk + B.ch[u]) * 4 + j) * 64 + lane) * 16);
          else if constexpr (FR) B.b[u][nb][j] = *reinterpret_cast<const u32x4*>(Wb + ((((int64_t)(kk * CBt + col0 / 32 + nb) * nchunk + B.ch[u]) * NJ + j) * 64 + lane) * 16);
          else if constexpr (X3) B.b[u][nb][j] = *reinterpret_cast<const u32x4*>(Wb + (((int64_t)kk * p.Cout + col0 + nb * 32 + fi) * p.Cin + B.ch[u] * 32) * EB +
                                                                                  ((j < 2 ? 2 * j : 4 + 2 * (j - 2)) + fh) * 16);
          else B.b[u][nb][j] = *reinterpret_cast<const u32x4*>(Wb + (((int64_t)kk * p.Cout + col0 + nb * 32 + fi) * p.Cin + B.ch[u] * 32) * EB + j * 32 + fh * 16);
    }
  };
  auto contract = [&](Batch& B) __attribute__((always_inline)) {
    if (pro) {                                                // gather-side BatchNorm+ReLU (module-by-module path only)
#pragma unroll
      for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (B.idx[u] < 0) continue;
          const int c0 = B.ch[u] * 32 + (j * 32 + fh * 16) / EB;
          if constexpr (BF16) {
            u32x4 v = B.a[u][j];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float lo = bf16_lo(v[q]), hi = bf16_hi(v[q]);
              if (p.in_scale) { lo = fmaf(lo, p.in_scale[c0 + 2 * q], p.in_shift[c0 + 2 * q]); hi = fmaf(hi, p.in_scale[c0 + 2 * q + 1], p.in_shift[c0 + 2 * q + 1]); }
              if (p.in_relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
              v[q] = pack_bf16x2(lo, hi);
            }
            B.a[u][j] = v;
          } else {
            f32x4 v = __builtin_bit_cast(f32x4, B.a[u][j]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (p.in_scale) v[q] = fmaf(v[q], p.in_scale[c0 + q], p.in_shift[c0 + q]);
              if (p.in_relu) v[q] = fmaxf(v[q], 0.f);
            }
            B.a[u][j] = __builtin_bit_cast(u32x4, v);
          }
        }
    }
    if constexpr (X3) {
#pragma unroll
      for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int J = 0; J < 2; ++J) {
          u32x4 ah, al;
          x3_split8(B.a[u][2 * J], B.a[u][2 * J + 1], ah, al);
#pragma unroll
          for (int nb = 0; nb < NBB; ++nb) mma16_x3(acc[nb], ah, al, B.b[u][nb][J], B.b[u][nb][2 + J]);
        }
    } else
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int nb = 0; nb < NBB; ++nb)
#pragma unroll
        for (int j = 0; j < NJ; ++j) mma16<BF16>(acc[nb], B.a[u][j], B.b[u][nb][j]);   // steps past the end hold zero A fragments
  };
  Batch B0;
  for (int s0 = wv; s0 < nsteps; s0 += WV * PF) { request(s0, B0); contract(B0); }

  // wave w finishes registers r = (16/WV) w .. of every column block (fixed summation order => deterministic); the column blocks
  // go through the reduction buffer one after the other (a full-width workgroup has up to 7 of them)
  const int col = lane & 31;
  constexpr int RPW = 16 / WV;
#pragma unroll
  for (int nb = 0; nb < NBB; ++nb) {
    if (nb > 0) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv][r][lane] = acc[nb][r];
    __syncthreads();
    const int j = col0 + nb * 32 + col;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = wv * RPW + rr;
      const int64_t orow = (int64_t)rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (orow >= p.n_out) continue;
      float v = red[0][r][lane];
#pragma unroll
      for (int w = 1; w < WV; ++w) v += red[w][r][lane];
      if constexpr (BF16) {
        if (p.res) v += bf16_lo((uint32_t)((const uint16_t*)p.res)[orow * p.res_ld + j]);
      } else {
        if (p.res) v += ((const float*)p.res)[orow * p.res_ld + j];
      }
      epi_store1<BF16>(p.out, p.out_ld, p.out_scale, p.out_shift, p.out_relu, orow, j, v);
      if (p.out2) epi_store1<BF16>(p.out2, p.out2_ld, p.out2_scale, p.out2_shift, p.out2_relu, orow, j, v);
      if (p.out3) epi_store1<BF16>(p.out3, p.out3_ld, p.out3_scale, p.out3_shift, p.out3_relu, orow, j, v);
    }
  }
}

// ------------------------------------------------------------------ Cin <= 8
template <typename T, int CIN>
__global__ void __launch_bounds__(256) k_conv_tinycin(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) float wsh[];        // [K][CIN][Cout]
  const T* w = (const T*)p.w; const T* in = (const T*)p.in; const T* res = (const T*)p.res;
  for (int e = threadIdx.x; e < p.K * CIN * p.Cout; e += 256) {
    const int j = e % p.Cout, c = (e / p.Cout) % CIN, k = e / (p.Cout * CIN);
    wsh[e] = ld_elem(w + ((int64_t)k * p.Cout + j) * p.Cin + c);
  }
  __syncthreads();
  const int ngrp = p.Cout / 8;                                         // 8 output channels per thread
  const int64_t total = p.n_out * ngrp;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t o = t / ngrp; const int j0 = (int)(t % ngrp) * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.K; ++k) {
      const int64_t idx = p.table ? (int64_t)p.table[(int64_t)k * p.n_out + o] : o;
      if (idx < 0) continue;
      float x[CIN];
#pragma unroll
      for (int c = 0; c < CIN; ++c) x[c] = ld_elem(in + idx * p.in_ld + c);
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        const float4 w0 = *reinterpret_cast<const float4*>(&wsh[(k * CIN + c) * p.Cout + j0]);
        const float4 w1 = *reinterpret_cast<const float4*>(&wsh[(k * CIN + c) * p.Cout + j0 + 4]);
        acc[0] = fmaf(x[c], w0.x, acc[0]); acc[1] = fmaf(x[c], w0.y, acc[1]); acc[2] = fmaf(x[c], w0.z, acc[2]); acc[3] = fmaf(x[c], w0.w, acc[3]);
        acc[4] = fmaf(x[c], w1.x, acc[4]); acc[5] = fmaf(x[c], w1.y, acc[5]); acc[6] = fmaf(x[c], w1.z, acc[6]); acc[7] = fmaf(x[c], w1.w, acc[7]);
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) if (res) acc[q] += ld_elem(res + o * p.res_ld + j0 + q);
    constexpr bool BF = sizeof(T) == 2;
    epi_store8<BF>(p.out, p.out_ld, p.out_scale, p.out_shift, p.out_relu, o, j0, acc);
    if (p.out2) epi_store8<BF>(p.out2, p.out2_ld, p.out2_scale, p.out2_shift, p.out2_relu, o, j0, acc);
    if (p.out3) epi_store8<BF>(p.out3, p.out3_ld, p.out3_scale, p.out3_shift, p.out3_relu, o, j0, acc);
  }
}

template <typename T>
int launch_tiny(const ConvP& p, hipStream_t s) {
  const size_t lds = (size_t)p.K * p.Cin * p.Cout * 4;
  const unsigned g = tl_grid(p.n_out * (p.Cout / 8), 256);
  switch (p.Cin) {
    case 1: k_conv_tinycin<T, 1><<<g, 256, lds, s>>>(p); break;
    case 2: k_conv_tinycin<T, 2><<<g, 256, lds, s>>>(p); break;
    case 3: k_conv_tinycin<T, 3><<<g, 256, lds, s>>>(p); break;
    case 4: k_conv_tinycin<T, 4><<<g, 256, lds, s>>>(p); break;
    case 5: k_conv_tinycin<T, 5><<<g, 256, lds, s>>>(p); break;
    case 6: k_conv_tinycin<T, 6><<<g, 256, lds, s>>>(p); break;
    case 7: k_conv_tinycin<T, 7><<<g, 256, lds, s>>>(p); break;
    case 8: k_conv_tinycin<T, 8><<<g, 256, lds, s>>>(p); break;
    default: return TL_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

}  // namespace

int g_small_mode = 0;   // developer A/B (tl_set_tuning "small_mode"): 0 = by size, 1 = always 4 waves (no full-width blocks), 2 = always 8 waves x 32 columns, 3 = ignore the fragment-order weights

int tl_launch_conv_small(const ConvP& p, int dtype, hipStream_t s) {
  const int nrt = (int)tl_cdiv(p.n_out, 32);
  // few row tiles: 8 waves split the (tap, chunk) steps of a 32 x 32 output block (more, shorter dependent chains);
  // otherwise 4 waves per 32 x 64 (or 32 x 32) block
  const bool eight = g_small_mode == 2 || (g_small_mode == 0 && nrt * (p.Cout / 32) <= 512 && p.K * (p.Cin / 32) >= 32);   // measured: l=6,7 1.5x, l=5 equal, 1x1 slower
  // full width (bf16, 96..224 output channels, enough row tiles to occupy the chip): ONE workgroup per 32-row tile computes every
  // column block, so the gathered rows are read once instead of once per column block (level 5 of config 2, 160 -> 160: the 980
  // 32 x 32 blocks pulled 541 MB through L2 per conv, which is what bounded it; 196 full-width blocks pull 160 MB)
  const int cbt = p.Cout / 32;
  // measured (tools/dev_small.py): 160 -> 160 at 6 264 rows 37.6 -> 31.1 us, 320 -> 160 82.6 -> 51.2, 96 -> 96 at 16 k rows 34.4 -> 26.8;
  // widths that are multiples of 64 already run 32 x 64 blocks and lose (128 -> 128: 34.6 -> 43.1 us), so: odd multiples of 32 only
  const bool full = g_small_mode == 0 && dtype == TL_BF16 && (cbt == 3 || cbt == 5 || cbt == 7) && nrt >= 128 && p.w_frag != nullptr &&
                    ((uintptr_t)p.w_frag) % 16 == 0;
  if (full) {
    switch (cbt) {
      case 3: k_conv_small<true, 3, 4, true><<<nrt, 256, 0, s>>>(p, 1); break;
      case 5: k_conv_small<true, 5, 4, true><<<nrt, 256, 0, s>>>(p, 1); break;
      case 7: k_conv_small<true, 7, 4, true><<<nrt, 256, 0, s>>>(p, 1); break;
    }
    return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
  }
  const bool two = !eight && (p.Cout % 64 == 0);
  const int ncb = p.Cout / (two ? 64 : 32);
  const unsigned g = (unsigned)(nrt * ncb);
#define TL_SMALL(BF_, FR_)                                                                   \
  do {                                                                                         \
    if (eight) k_conv_small<BF_, 1, 8, FR_><<<g, 512, 0, s>>>(p, ncb);                         \
    else if (two) k_conv_small<BF_, 2, 4, FR_><<<g, 256, 0, s>>>(p, ncb);                      \
    else k_conv_small<BF_, 1, 4, FR_><<<g, 256, 0, s>>>(p, ncb);                               \
  } while (0)
  const bool fr = p.w_frag != nullptr && ((uintptr_t)p.w_frag) % 16 == 0 && g_small_mode != 3;
  // fp32 rows with split-bf16 weights (the bf16x3 mode; weights of >= 256 input channels come as two half-width convs: exact kernel)
  const bool x3 = dtype == TL_F32 && p.w_x3 != nullptr && p.Cin < 256 && !p.in_scale && !p.in_relu;
  if (dtype == TL_BF16) { if (fr) TL_SMALL(true, true); else TL_SMALL(true, false); }
  else if (x3 && g_small_mode != 3 && p.Cout % 32 == 0) {                        // (fragment-order copy of the split weights: 1 KB per load instead of 64 x 16 B)
    if (eight) k_conv_small<false, 1, 8, true, true><<<g, 512, 0, s>>>(p, ncb);
    else if (two) k_conv_small<false, 2, 4, true, true><<<g, 256, 0, s>>>(p, ncb);
    else k_conv_small<false, 1, 4, true, true><<<g, 256, 0, s>>>(p, ncb);
  }
  else if (x3) {
    if (eight) k_conv_small<false, 1, 8, false, true><<<g, 512, 0, s>>>(p, ncb);
    else if (two) k_conv_small<false, 2, 4, false, true><<<g, 256, 0, s>>>(p, ncb);
    else k_conv_small<false, 1, 4, false, true><<<g, 256, 0, s>>>(p, ncb);
  }
  else { if (fr) TL_SMALL(false, true); else TL_SMALL(false, false); }
#undef TL_SMALL
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

int tl_launch_conv_tinycin(const ConvP& p, int dtype, hipStream_t s) {
  const bool out_ok = (dtype == TL_F32) ? (p.out_ld % 4 == 0 && ((uintptr_t)p.out) % 16 == 0) : (p.out_ld % 8 == 0 && ((uintptr_t)p.out) % 16 == 0);
  if (!out_ok) return TL_ERR_ARG;
  return dtype == TL_F32 ? launch_tiny<float>(p, s) : launch_tiny<__hip_bfloat16>(p, s);
}
