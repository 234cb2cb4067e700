// Sparse conv, "stream-q" form (bf16, Cin a multiple of 64): tl_conv_stream.hip with coalesced gathers.
//
// Why: a 32x32x16 MFMA wants its A operand as (lane & 31) = row, (lane >> 5) = which 16 B of the row -- loading that shape
// straight from global memory makes every lane of a buffer_load_b128 its own 16-B request (32 rows x 2 pieces per
// instruction).  Measured on the level-2 rulebook (tools/dev_gather.py): 27 taps of such gathers alone take 0.26 ms, while
// the same bytes fetched with four ADJACENT lanes reading 64 contiguous bytes take 0.135 ms -- the texture-address path
// works on quads.  And tools/dev_stream_tm.py showed the stream kernel's waves stalled a quarter of their time just
// issuing the fragment-shaped loads (the request queue was full) and another third at the per-tap barrier behind them.
//
// So here a wave gathers its 32 rows quad-wise -- load i (0..3) of a 128-B part: lane l reads row ((l>>2)&7) + 8 i, bytes
// 64 (l>>5) + 16 (l&3) .. +16, i.e. 8 whole rows per instruction -- and then turns (load index) x (lane & 3) around in
// registers: two butterfly stages of v_cndmask + quad-permute DPP (lane bit 0 <-> load bit 0, lane bit 1 <-> load bit 1).
// Afterwards register s (0..3) of lane l holds 16-B piece s + 4 (l>>5) of tile row rho(l & 31) = ((l>>2)&7) + 8 (l&3):
// a valid A operand for k-step s if (1) the weight fragment of k-step s, half h is read from piece s + 4 h (free: the
// order of the contraction index is ours) and (2) MFMA row m is written back as tile row rho(m) in the epilogue.
// The VALU work (32 ops per 128-B part and tap) rides on the otherwise idle vector ALU next to the MFMAs.
//
// Row indices: each tap needs 4 per lane now (rows q + 8 i), so the wave's [K][32] slice of the rulebook is parked in LDS once
// (as [K][8][4] -> one broadcast ds_read_b128 per tap) instead of 27 registers per lane.
// Weight streaming, double buffering, request order (weights of tap t at step t - WA before that step's gathers) and the
// epilogue are those of tl_conv_stream.hip.  Deterministic; all K taps contracted.
#include "tl_conv_internal.h"
#include <atomic>
#include <type_traits>

namespace {

__device__ unsigned long long g_tmq[8];   // developer timing mode (tl_dev_streamq_tm)

template <int CTRL>
static __device__ __forceinline__ uint32_t qperm(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}

// S[i] (i = load index) -> F[s]: see the header.  0xB1 = quad_perm [1,0,3,2] (lane ^ 1), 0x4E = [2,3,0,1] (lane ^ 2)
static __device__ __forceinline__ void quad_transpose(const u32x4 (&S)[4], u32x4 (&F)[4], bool o0, bool o1) {
  u32x4 T[4];                                      // (the permutes are computed unconditionally: DPP must see all lanes)
#pragma unroll
  for (int i1 = 0; i1 < 2; ++i1)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t P = S[2 * i1][d], Q = S[2 * i1 + 1][d];
      const uint32_t Pn = qperm<0xB1>(P), Qn = qperm<0xB1>(Q);
      T[2 * i1][d] = o0 ? Qn : P;
      T[2 * i1 + 1][d] = o0 ? Q : Pn;
    }
#pragma unroll
  for (int x0 = 0; x0 < 2; ++x0)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t Ta = T[x0][d], Tb = T[2 + x0][d];
      const uint32_t An = qperm<0x4E>(Ta), Bn = qperm<0x4E>(Tb);
      F[x0][d] = o1 ? Bn : Ta;
      F[x0 + 2][d] = o1 ? Tb : An;
    }
}

// W waves per workgroup, each owning RB blocks of 32 rows (one weight fragment read from LDS feeds RB MFMAs).
// ABL: developer ablation bits (1 no transposition, 2 no gathers, 4 no barrier, 8 no MFMA, 16 timers, 32 MFMAs for taps < 16 only)
// SP: the input channels are walked in SP slices of PN * 64: step v of the K * SP steps contracts slice v % SP of tap v / SP
// (same registers and LDS as the PN-wide kernel; 256 -> 128 as SP = 2 x 128 instead of a PN = 4 kernel that spills).
// X3 (round 6; fp32 rows, the parity-fast mode "bf16x3"): a 128-B part is 32 fp32 channels.  The gathers and the transposition do not care --
// afterwards register s of lane half h holds the fp32 channels 16 h + 4 s .. + 3 of the part --, and the pairs (s, s + 2) = (t, t + 2) are
// exactly the two 16-B pieces x3_split8 turns into the hi / lo operand of weight slot (J = h, fh = t) of tl_pack_weight_x3's layout
// (channels 16 J + 4 fh + {0..3} and 16 J + 8 + 4 fh + {0..3}), whose rows [K][Cout][Cin / 32][128 B] are byte for byte the weight rows this
// kernel streams: two k-steps per part, three MFMAs each (lo.Whi + hi.Wlo + hi.Whi), fp32 epilogue.  The split-bf16 form of tl_conv_stream.hip
// issues one 16-B request per lane and piece (fragment shape); here four adjacent lanes read 64 contiguous bytes.
template <int K, int NB, int PN, int DA, int W, int RB, int OCC, int ABL, int SP = 1, bool X3 = false, bool UL = false>
__global__ void __launch_bounds__(W * 64, OCC) k_conv_streamq(ConvP p) {
  constexpr bool TM = (ABL & 16) != 0;
  constexpr int KV = K * SP;
  constexpr int NTH = W * 64;
  constexpr int COUT = NB * 32, CIN = PN * 64;
  constexpr int BROW = CIN * 2 + 16;                  // LDS pitch of a weight row (+16 B: conflict-free ds_read_b128 down a column)
  constexpr int BSLOTS = PN * 8;                      // 16-B pieces per weight row
  constexpr int BVEC = COUT * BSLOTS;
  constexpr int BPT = (BVEC + NTH - 1) / NTH;
  constexpr int EP = 32 + 4;
  constexpr int WB = 2 * COUT * BROW;
  constexpr int LA = 4 * PN * RB;                     // gather instructions per tap
  constexpr int WROWS = 32 * RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Bs = smem;                                                          // [2][COUT][BROW]
  int* Is = reinterpret_cast<int*>(smem + WB);                              // [W][K][RB][8][4] row indices
  float* Es = reinterpret_cast<float*>(smem);                               // epilogue alias: [W][32][EP]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  const bool o0 = lane & 1, o1 = lane & 2;
  const int tile = xcd_tile(blockIdx.x, p.nblk);
  const int64_t r0 = (int64_t)tile * (W * WROWS) + wv * WROWS;

  // the wave's slice of the rulebook -> LDS, [k][rb][row & 7][row >> 3]
  int* iw = Is + wv * (K * WROWS);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int64_t row = r0 + rb * 32 + fi;
    const bool rvalid = row < p.n_out;
    int v[(K + 1) / 2];
    if (K == 27 && p.ctab) {                                 // column form of the rulebook: 40 B per row instead of 108
      int t27[27];
      decode_ctab(p.ctab, p.n_out, row, rvalid, t27);
#pragma unroll
      for (int t = 0; t < (K + 1) / 2; ++t) v[t] = fh ? t27[min(2 * t + 1, 26)] : t27[min(2 * t, 26)];
    } else {
#pragma unroll
      for (int t = 0; t < (K + 1) / 2; ++t) {
        const int k = min(2 * t + fh, K - 1);
        v[t] = rvalid ? (p.table ? p.table[(int64_t)k * p.n_out + row] : (int)row) : -1;
      }
    }
#pragma unroll
    for (int t = 0; t < (K + 1) / 2; ++t) {
      const int k = 2 * t + fh;
      if (k < K) iw[((k * RB + rb) * 8 + (fi & 7)) * 4 + (fi >> 3)] = v[t];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  const int in_ld_b = (int)(p.in_ld * (X3 ? 4 : 2));
  const int64_t in_bytes = ((int64_t)p.n_in - 1) * in_ld_b + (int64_t)CIN * SP * 2;          // (CIN * 2 = PN * 128 bytes per slice in either mode)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)in_bytes, 0x00020000);
  const unsigned qoff = (unsigned)(fh * 64 + (lane & 3) * 16);
  const int* iq = iw + ((lane >> 2) & 7) * 4;
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);

  f32x16 acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][nb][i] = 0.f;

  constexpr int WA = DA > 2 ? DA : 2, RW = WA - 1;
  u32x4 a[DA][RB][PN][4];
  u32x4 bw[RW][BPT], bw0[BPT];
  u32x4 idn[RB];                                       // row indices of the next tap to request
  auto read_idx = [&](int v) __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) idn[rb] = *reinterpret_cast<const u32x4*>(iq + ((v / SP) * RB + rb) * 32);
  };
  auto issue_a = [&](int v, u32x4 (&dst)[RB][PN][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned base = idn[rb][i] * (unsigned)in_ld_b + qoff + (unsigned)((v % SP) * CIN * 2);
#pragma unroll
        for (int pp = 0; pp < PN; ++pp) {
          if constexpr (ABL & 2) dst[rb][pp][i] = u32x4{base, base, base, base};
          else dst[rb][pp][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + pp * 128), 0, 0));
        }
      }
  };
  auto load_b = [&](int kv, u32x4 (&dst)[BPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const int v = BVEC % NTH == 0 ? tid + q * NTH : min(tid + q * NTH, BVEC - 1);
      if constexpr (SP == 1) dst[q] = wsrc[(int64_t)kv * BVEC + v];
      else dst[q] = wsrc[(((int64_t)(kv / SP) * COUT + v / BSLOTS) * SP + kv % SP) * BSLOTS + v % BSLOTS];   // [K][COUT][SP][BSLOTS]
    }
  };
  auto store_b = [&](int buf, const u32x4 (&src)[BPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const int v = tid + q * NTH;
      const int n = v / BSLOTS, s = v % BSLOTS;
      if (BVEC % NTH == 0 || v < BVEC) *reinterpret_cast<u32x4*>(Bs + buf * COUT * BROW + n * BROW + s * 16) = src[q];
    }
  };

  load_b(0, bw0);
  read_idx(0);
#pragma unroll
  for (int d = 0; d < WA; ++d) {
    if (d >= 1 && d < KV) load_b(d, bw[d % RW]);
    if (d < DA && d < KV) { issue_a(d, a[d]); if (d + 1 < KV) read_idx(d + 1); }
  }
  store_b(0, bw0);
  __syncthreads();

  [[maybe_unused]] unsigned long long tm[4] = {0, 0, 0, 0}, tprev = 0;
  auto tick = [&](int seg) __attribute__((always_inline)) {
    if constexpr (TM) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (seg >= 0) tm[seg] += t - tprev;
      tprev = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  tick(-1);
  // one step of the tap loop; U = k % DA (and the parity of k: DA is even) as a compile-time constant, so that the register arrays keep
  // static indices when the loop is NOT fully unrolled (the X3 bodies are three times the size of the bf16 ones: hipcc refuses to unroll
  // 27 / 54 of them and would otherwise index a[] / bw[] dynamically, i.e. through scratch memory)
  auto step = [&](int k, auto Uc) __attribute__((always_inline)) {
    constexpr int U = decltype(Uc)::value;
    if (k + 1 < KV) store_b((U + 1) & 1, bw[(U + 1) % RW]);
    if (k + WA < KV) load_b(k + WA, bw[(U + WA) % RW]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TM) { if (k + DA < KV) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DA - 1) * (LA + BPT) + BPT)); else asm volatile("s_waitcnt vmcnt(0)"); }
    tick(0);
    u32x4 F[RB][PN][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int pp = 0; pp < PN; ++pp) {
        if constexpr (ABL & 1) { for (int i = 0; i < 4; ++i) F[rb][pp][i] = a[U % DA][rb][pp][i]; }
        else quad_transpose(a[U % DA][rb][pp], F[rb][pp], o0, o1);
      }
    const char* bl = Bs + (U & 1) * COUT * BROW + fi * BROW + fh * 64;
    if constexpr (X3) {
#pragma unroll
      for (int pp = 0; pp < PN; ++pp)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          u32x4 ah[RB], al[RB];
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) x3_split8(F[rb][pp][t], F[rb][pp][2 + t], ah[rb], al[rb]);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const char* wb = bl - fh * 64 + nb * 32 * BROW + pp * 128;                      // the unit's 128 B: hi slots 2 J + fh, lo slots 4 + 2 J + fh (J = lane half)
            const u32x4 bh = *reinterpret_cast<const u32x4*>(wb + (2 * fh + t) * 16);
            const u32x4 blo = *reinterpret_cast<const u32x4*>(wb + (4 + 2 * fh + t) * 16);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) mma16_x3(acc[rb][nb], ah[rb], al[rb], bh, blo);
          }
        }
    } else
#pragma unroll
    for (int pp = 0; pp < PN; ++pp)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const u32x4 bf = *reinterpret_cast<const u32x4*>(bl + nb * 32 * BROW + (pp * 8 + s) * 16);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            if constexpr ((ABL & 8) != 0) acc[rb][nb][0] += __uint_as_float(F[rb][pp][s][0] ^ bf[0]);
            else if ((ABL & 32) != 0 && k / SP >= 16) acc[rb][nb][0] += __uint_as_float(F[rb][pp][s][0] ^ bf[0]);
            else mma16<true>(acc[rb][nb], F[rb][pp][s], bf);
          }
        }
    tick(1);
    if (k + DA < KV) { issue_a(k + DA, a[U % DA]); if (k + DA + 1 < KV) read_idx(k + DA + 1); }
    tick(2);
    if constexpr ((ABL & 4) == 0) { if (k + 1 < KV) __syncthreads(); }
    tick(3);
  };
  if constexpr (X3 && !UL) {                        // two steps per trip: the parity of k (and with it every register-array index) stays static
    for (int k0 = 0; k0 < KV; k0 += 2) {
      step(k0, std::integral_constant<int, 0>{});
      if (k0 + 1 < KV) step(k0 + 1, std::integral_constant<int, 1>{});
    }
  } else if constexpr (X3) {                          // UL: fully unrolled (the instantiations small enough for hipcc to do it without spilling)
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      if ((k & 1) == 0) step(k, std::integral_constant<int, 0>{});
      else step(k, std::integral_constant<int, 1>{});
    }
  } else {
    // (the 16-bit instantiations keep the fully unrolled loop they were tuned with)
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    if (k + 1 < KV) store_b((k + 1) & 1, bw[(k + 1) % RW]);
    if (k + WA < KV) load_b(k + WA, bw[(k + WA) % RW]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TM) { if (k + DA < KV) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DA - 1) * (LA + BPT) + BPT)); else asm volatile("s_waitcnt vmcnt(0)"); }
    tick(0);
    u32x4 F[RB][PN][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int pp = 0; pp < PN; ++pp) {
        if constexpr (ABL & 1) { for (int i = 0; i < 4; ++i) F[rb][pp][i] = a[k % DA][rb][pp][i]; }
        else quad_transpose(a[k % DA][rb][pp], F[rb][pp], o0, o1);
      }
    const char* bl = Bs + (k & 1) * COUT * BROW + fi * BROW + fh * 64;
#pragma unroll
    for (int pp = 0; pp < PN; ++pp)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const u32x4 bf = *reinterpret_cast<const u32x4*>(bl + nb * 32 * BROW + (pp * 8 + s) * 16);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            if constexpr ((ABL & 8) != 0) acc[rb][nb][0] += __uint_as_float(F[rb][pp][s][0] ^ bf[0]);
            else if ((ABL & 32) != 0 && k / SP >= 16) acc[rb][nb][0] += __uint_as_float(F[rb][pp][s][0] ^ bf[0]);
            else mma16<true>(acc[rb][nb], F[rb][pp][s], bf);
          }
        }
    tick(1);
    if (k + DA < KV) { issue_a(k + DA, a[k % DA]); if (k + DA + 1 < KV) read_idx(k + DA + 1); }
    tick(2);
    if constexpr ((ABL & 4) == 0) { if (k + 1 < KV) __syncthreads(); }
    tick(3);
  }
  }
  if constexpr (TM) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) atomicAdd(&g_tmq[i], tm[i]);
      atomicAdd(&g_tmq[4], 1ull);
    }
  }

  // epilogue: MFMA row m = (r & 3) + 8 (r >> 2) + 4 fh is tile row rho(m) = ((m >> 2) & 7) + 8 (m & 3) = 2 (r >> 2) + fh + 8 (r & 3)
  __syncthreads();
  float* ew = Es + wv * 32 * EP;
  float red0[NB], red1[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) { red0[nb] = 0.f; red1[nb] = 0.f; }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ew[(2 * (r >> 2) + fh + 8 * (r & 3)) * EP + fi] = acc[rb][nb][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      epi_block32<!X3, EP>(p, ew, lane, r0 + rb * 32, nb * 32, red0[nb], red1[nb]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  if (p.epi_mode != TL_EPI_NONE) epi_finish_wg<W, EP, NB>(p, Es, tid, red0, red1);
}

template <int K, int NB, int PN, int DA, int W = 8, int RB = 1, int ABL = 0, int SP = 1, bool X3 = false, bool UL = false>
int launch(ConvP p, hipStream_t s) {
  constexpr int OCC = 2;
  if constexpr (X3) p.w = p.w_x3;
  const size_t wt = 2 * (size_t)NB * 32 * (PN * 128 + 16) + (size_t)W * K * 32 * RB * 4, ep = (size_t)W * 32 * 36 * 4;
  const size_t lds = wt > ep ? wt : ep;
  if (lds > 160 * 1024) return TL_ERR_UNSUPPORTED;
  static TlAttrOnce attr_once;                     // per kernel instantiation AND device (the attribute is per device)
  if (!tl_lds_attr(attr_once, reinterpret_cast<const void*>(&k_conv_streamq<K, NB, PN, DA, W, RB, OCC, ABL, SP, X3, UL>), 160 * 1024)) return TL_ERR_LAUNCH;
  p.nblk = (int)tl_cdiv(p.n_out, W * 32 * RB);
  k_conv_streamq<K, NB, PN, DA, W, RB, OCC, ABL, SP, X3, UL><<<p.nblk, W * 64, lds, s>>>(p);
  if (p.red_nparts) *p.red_nparts = p.nblk;
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

#ifdef TL_DEV
int g_tm_on = 0;
#endif

}  // namespace

#ifdef TL_DEV
// Developer hook (dev build only, `python -m treelearn_amd.build --dev`; not part of the C ABI): per-segment cycle counters of
// the 64->64 shape on/off, read and clear.
extern "C" int tl_dev_streamq_tm(int enable, unsigned long long* out8) {
  g_tm_on = enable;
  if (out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tmq), sizeof(g_tmq)) != hipSuccess) return TL_ERR_LAUNCH;
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tmq), z, sizeof(z)) != hipSuccess) return TL_ERR_LAUNCH;
  }
  return TL_OK;
}
#endif

int tl_launch_conv_streamq(const ConvP& p, hipStream_t s) {
  if (p.in_scale || p.in_relu || p.Cin % 64 || p.Cout % 32) return TL_ERR_UNSUPPORTED;
  const int64_t ld_b = p.in_ld * 2, in_bytes = (p.n_in - 1) * ld_b + (int64_t)p.Cin * 2;
  if (!(in_bytes > 0 && in_bytes + 2 * ld_b < 0xFFFFFFFFll)) return TL_ERR_UNSUPPORTED;
  const int nb = p.Cout / 32, pn = p.Cin / 64;
  if (p.K == 27) {
#ifdef TL_DEV
    if (g_tm_on && nb == 2 && pn == 1) {
      switch (g_tm_on) {
        case 1: return launch<27, 2, 1, 2, 8, 1, 16>(p, s);
        case 2: return launch<27, 2, 1, 2, 8, 1, 1>(p, s);
        case 3: return launch<27, 2, 1, 2, 8, 1, 2>(p, s);
        case 4: return launch<27, 2, 1, 2, 8, 1, 4>(p, s);
        case 6: return launch<27, 2, 1, 2, 8, 1, 3>(p, s);
        case 8: return launch<27, 2, 1, 3, 8, 1>(p, s);
        case 9: return launch<27, 2, 1, 1, 8, 1>(p, s);
        case 10: return launch<27, 2, 1, 2, 4, 1>(p, s);
        case 11: return launch<27, 2, 1, 2, 4, 2>(p, s);
        case 12: return launch<27, 2, 1, 2, 8, 2>(p, s);
        case 13: return launch<27, 2, 1, 1, 4, 2>(p, s);
        case 14: return launch<27, 2, 1, 3, 4, 2>(p, s);
        case 15: return launch<27, 2, 1, 2, 4, 2, 2>(p, s);
        case 20: return p.table ? launch<16, 2, 1, 2>(p, s) : TL_ERR_ARG;      // the first 16 taps only: gathers AND MFMAs (tools/dev_l2_floor.py)
        case 21: return launch<27, 2, 1, 2, 8, 1, 32>(p, s);                   // all 27 gathers, MFMAs for 16 taps
      }
    }
#endif
    if (nb == 2 && pn == 1) return launch<27, 2, 1, 2>(p, s);
    if (nb == 2 && pn == 2) return launch<27, 2, 2, 2>(p, s);
    if (nb == 4 && pn == 2) return launch<27, 4, 2, 2>(p, s);
    if (nb == 4 && pn == 1) return launch<27, 4, 1, 2>(p, s);          // 64 -> 128: the dgrad of the level-2 128 -> 64 conv
    if (nb == 3 && pn == 3) return launch<27, 3, 3, 2>(p, s);
    if (nb == 4 && pn == 4) return launch<27, 4, 2, 2, 8, 1, 0, 2>(p, s);
  } else if (p.K == 8) {
    if (nb == 3 && pn == 1) return launch<8, 3, 1, 2>(p, s);
  }
  return TL_ERR_UNSUPPORTED;
}

#ifndef TL_F16_BUILD
// fp32 rows, split-bf16 contraction (p.w_x3 in the tl_pack_weight_x3 layout), 27 taps, Cin a multiple of 32 (one 128-B part per 32 channels)
int tl_launch_conv_streamq_x3(const ConvP& p, int mode, hipStream_t s) {
  if (!p.w_x3 || p.in_scale || p.in_relu || p.Cin % 32 || p.Cout % 32 || p.K != 27 || p.epi_mode != TL_EPI_NONE || p.Cin >= 256) return TL_ERR_UNSUPPORTED;
  const int64_t ld_b = p.in_ld * 4, in_bytes = (p.n_in - 1) * ld_b + (int64_t)p.Cin * 4;
  if (!(in_bytes > 0 && in_bytes + 2 * ld_b < 0xFFFFFFFFll) || ((uintptr_t)p.w_x3) % 16) return TL_ERR_UNSUPPORTED;
  const int nb = p.Cout / 32, pn = p.Cin / 32;
  // developer modes (tl_set_tuning "streamq_x3"): 0 = off, 1 = the shipped choice, 2 = prefetch depth 2 / two-step loop for every shape, 3 = depth 1 /
  // two-step loop for every shape
  if (mode == 2) {
    if (nb == 2 && pn == 2) return launch<27, 2, 2, 2, 8, 1, 0, 1, true>(p, s);
    if (nb == 2 && pn == 4) return launch<27, 2, 2, 2, 8, 1, 0, 2, true>(p, s);
    if (nb == 3 && pn == 3) return launch<27, 3, 3, 2, 8, 1, 0, 1, true>(p, s);
    if (nb == 3 && pn == 6) return launch<27, 3, 3, 2, 8, 1, 0, 2, true>(p, s);
  }
  if (mode == 3) {
    if (nb == 2 && pn == 2) return launch<27, 2, 2, 1, 8, 1, 0, 1, true>(p, s);
    if (nb == 2 && pn == 4) return launch<27, 2, 2, 1, 8, 1, 0, 2, true>(p, s);
    if (nb == 3 && pn == 3) return launch<27, 3, 3, 1, 8, 1, 0, 1, true>(p, s);
    if (nb == 3 && pn == 6) return launch<27, 3, 3, 1, 8, 1, 0, 2, true>(p, s);
  }
  // shipped (config-2 rulebooks, profiles/r6_x3/x3_l2.txt; fragment-shape kernel -> this one): the 96 -> 96 shape gains nothing and stays there
  if (nb == 2 && pn == 2) return launch<27, 2, 2, 1, 8, 1, 0, 1, true, true>(p, s);    // 64 -> 64  (level 2): 0.79-0.88 -> 0.74-0.82 ms (depth 1, fully unrolled: 114 registers, two workgroups per CU)
  if (nb == 2 && pn == 4) return launch<27, 2, 2, 1, 8, 1, 0, 2, true>(p, s);          // 128 -> 64 (level 2 decoder): 1.75 -> 1.52 ms
  if (nb == 3 && pn == 6) return launch<27, 3, 3, 2, 8, 1, 0, 2, true>(p, s);          // 192 -> 96 (level 3 decoder): 0.89 -> 0.82 ms
  return TL_ERR_UNSUPPORTED;
}
#endif
