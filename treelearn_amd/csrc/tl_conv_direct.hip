// Sparse conv, "direct" form: for layers whose whole weight tensor fits in LDS (K*Cout*Cin*sizeof <= ~110 KB:
// every level-1 conv of the U-Net, the level-1/2 down/up convs and 1x1s).  bf16 and fp32.
//
// Measured on MI355X (tools/dev_gather.py): MFMA-fragment-shaped buffer gathers (32 rows x 32 B per instruction)
// of 64-B rows run as fast as the 8-lanes-per-row staging pattern (all 27 taps of the level-1 rulebook in 0.13 ms),
// while the LDS-staged tile kernel spends most of its time on LDS round trips and one barrier per step.  So here:
//   * all K taps of the weights are staged ONCE per workgroup in LDS (XOR-swizzled rows, conflict-free
//     ds_read_b128 B-fragments) and the workgroup walks many 32-row tiles (persistent waves);
//   * each WAVE owns a 32-row output tile end to end: A fragments are gathered straight from global memory into
//     MFMA operand registers with buffer loads (absent neighbour = index -1 = out-of-range offset = hardware
//     returns zeros: no masks, no clamps, no branches), taps grouped G at a time and double-buffered in registers;
//   * no barrier and no LDS traffic for A in the main loop; code is straight-line, so hipcc keeps counted vmcnt;
//   * all K taps are contracted (no tap skipping): at 5.5 of 27 present neighbours the bf16 MFMA work is still
//     < 10 % of the kernel time, and skipping would need divergent control flow around the loads.
// Deterministic (fixed summation order).  Epilogue: LDS transposition, 16-B stores, residual, up to three views.
#include "tl_conv_internal.h"
#include <atomic>

namespace {

// ABL: developer ablation bits (tools/dev_direct_abl.py; results are wrong on purpose): 1 no gathers, 2 no MFMA,
// 4 no output stores, 8 no rulebook loads (identity rows)
// CT: the rulebook comes in column form (p.ctab, 40 B per voxel; decode_ctab) instead of the 27-entry table (108 B)
__device__ unsigned long long g_tmd[8];   // developer timing mode (ABL bit 16): cycles summed over waves per tile segment

// TR: the training-mode epilogue (tl_conv_args.epi_mode) is compiled in; the inference instantiations (TR = false) carry none of it
// OH: every output row has at most ONE valid table entry (inverse conv): that row is gathered once and routed to its tap by a per-lane
// select (K gathers of which K - 1 fetch nothing otherwise); all K taps are still contracted, against zeros except one
// X3 (fp32 storage only): split-bf16 contraction (tl_conv_internal.h: mma16_x3) on weights in the tl_pack_weight_x3 form
template <bool BF16, int K, int NB, int UN, int G, int WAVES, int ABL = 0, bool CT = false, bool TR = false, bool OH = false, bool X3 = false>
__global__ void __launch_bounds__(WAVES * 64) k_conv_direct(ConvP p, int ntiles, int walk) {
  static_assert(!X3 || (!BF16 && !TR), "the split-bf16 contraction: fp32 rows, inference");
  constexpr bool TM = (ABL & 16) != 0;
  [[maybe_unused]] unsigned long long tm[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
  auto tick = [&](int seg) __attribute__((always_inline)) {
    if constexpr (TM) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (seg >= 0) tm[seg] += t - tprev;
      tprev = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  constexpr int EB = BF16 ? 2 : 4;                   // bytes per element
  constexpr int UB = 32 * EB;                        // bytes of one 32-channel unit of a row
  constexpr int NJ = UB / 32;                        // 16-B fragment pairs per unit (lane half h takes bytes j*32 + h*16)
  constexpr int SLOTS = UB / 16;
  constexpr int COUT = NB * 32;
  constexpr int NG = (K + G - 1) / G;
  constexpr int EP = COUT + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ws = smem;                                                          // [K][UN][COUT][UB], 16-B slots swizzled
  float* Es = reinterpret_cast<float*>(smem + (size_t)K * UN * COUT * UB);  // [WAVES][32][EP]
  double* Ds = reinterpret_cast<double*>(smem + (size_t)K * UN * COUT * UB + (size_t)WAVES * 32 * EP * 4);   // training mode only: [WAVES][2][COUT] column sums

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  auto swz_of = [](int n) { return BF16 ? ((n >> 2) & 3) : ((n >> 1) & 7); };   // conflict-free b128 reads for 64-B / 128-B rows

  {  // stage every tap of the weights (global layout [K][Cout][Cin]) into LDS
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);
    constexpr int NVEC = K * COUT * UN * SLOTS;
    for (int v = tid; v < NVEC; v += WAVES * 64) {
      const int s = v % SLOTS, c = (v / SLOTS) % UN, n = (v / (SLOTS * UN)) % COUT, k = v / (SLOTS * UN * COUT);
      *reinterpret_cast<u32x4*>(Ws + ((k * UN + c) * COUT + n) * UB + ((s ^ swz_of(n)) * 16)) = wsrc[v];
    }
  }
  __syncthreads();

  const int in_ld_b = (int)(p.in_ld * EB);
  const int64_t in_bytes = ((int64_t)p.n_in - 1) * in_ld_b + (int64_t)UN * UB;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)in_bytes, 0x00020000);
  const unsigned lane_off = (unsigned)(fh * 16);
  const int swz = swz_of(fi);
  const char* wl = Ws + fi * UB;
  float* ew = Es + wv * 32 * EP;
  [[maybe_unused]] double* dw = Ds + wv * 2 * COUT;
  if constexpr (TR) {
    for (int e = lane; e < 2 * COUT; e += 64) dw[e] = 0.0;
  }

  // walk 1 (developer A/B): every XCD (block b runs on XCD b % 8) walks its own contiguous eighth of the tiles
  const int nblk = walk ? ((int)gridDim.x >> 3) : (int)gridDim.x, bidx = walk ? ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int t8 = walk ? (((ntiles + 7) / 8 + WAVES - 1) / WAVES) * WAVES : ntiles;
  const int tbase = walk ? ((int)blockIdx.x & 7) * t8 : 0;
  // Column-form kernel: the lane's ten rulebook words of tile t+1 are requested right after the LAST gather group of tile t
  // (younger than every gather, so no gather wait ever includes them -- vmcnt retires in order) and are there when the next
  // tile starts; taps are decoded from the words on use instead of being kept as 27 indices.  The per-segment timers (ABL 16)
  // had shown a wave spending 24 % of a tile waiting for its rulebook entries; measured 0.200 -> 0.195 / 0.191 -> 0.175 ms.
  // Requesting the tile's residual vectors ahead as well (23 % of a tile is the epilogue) pushed the kernel past 128 VGPRs
  // and gave nothing (0.197 / 0.179); the 27-entry table form spills with any of this and keeps the plain order.
  constexpr bool PF = (K == 27 && CT) && (ABL & 32) == 0;
  constexpr int NW = (K == 27 && CT) ? 10 : K;
  auto load_words = [&](int tile_, int (&w)[NW]) __attribute__((always_inline)) {
    const int64_t row = (int64_t)tile_ * 32 + fi;
    const bool rvalid = tile_ < ntiles && row < p.n_out;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      if constexpr (K == 27 && CT) w[k] = rvalid ? p.ctab[(int64_t)k * p.n_out + row] : (k < 9 ? -1 : 0);
      else w[k] = rvalid ? ((p.table && !(ABL & 8)) ? p.table[(int64_t)k * p.n_out + row] : (int)row) : -1;
    }
  };
  // column form: tap k = 3 c + d of the lane's row from the column base w[c] and the presence bits (see decode_ctab)
  auto word_tap = [&](const int (&w)[NW], int k) __attribute__((always_inline)) {
    const uint32_t m = ((uint32_t)w[NW - 1] >> (3 * (k / 3))) & 7u;
    const int d = k % 3;
    return ((m >> d) & 1u) ? w[k / 3] + __builtin_popcount(m & ((1u << d) - 1u)) : -1;
  };
  constexpr int VROW = NB * 4;
  [[maybe_unused]] int twn[NW];
  if constexpr (PF) load_words(tbase + bidx * WAVES + wv, twn);
  for (int lt = bidx * WAVES + wv; lt < t8; lt += nblk * WAVES) {
    const int tile = tbase + lt;
    if (tile >= ntiles) break;
    tick(-1);
    [[maybe_unused]] int idx[PF ? 1 : K];
    [[maybe_unused]] int twc[NW];                              // PF: the lane's rulebook words (requested during the previous tile); taps are decoded on use
    if constexpr (PF) {
#pragma unroll
      for (int k = 0; k < NW; ++k) twc[k] = twn[k];
    } else {
      const int64_t row = (int64_t)tile * 32 + fi;
      const bool rvalid = row < p.n_out;
      if constexpr (K == 27 && CT) decode_ctab(p.ctab, p.n_out, row, rvalid, idx);
      else if (OH && p.one_hot == 2) {                            // packed one-hot table: ONE word per row = (input row << 3) | tap
        const int v = rvalid ? p.table[row] : -1;
#pragma unroll
        for (int k = 0; k < K; ++k) idx[k] = (v >= 0 && (v & 7) == k) ? (v >> 3) : -1;
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k) idx[k] = rvalid ? ((p.table && !(ABL & 8)) ? p.table[(int64_t)k * p.n_out + row] : (int)row) : -1;
      }
    }

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

    u32x4 a[2][G][UN][NJ];
    auto issue = [&](int g, u32x4 (&dst)[G][UN][NJ]) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < G; ++t) {
        const int k = g * G + t;
        if (k < K) {
          int ik;
          if constexpr (PF) ik = word_tap(twc, k); else ik = idx[k];
          const unsigned base = (unsigned)ik * (unsigned)in_ld_b + lane_off;
#pragma unroll
          for (int c = 0; c < UN; ++c)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              if constexpr (ABL & 1) dst[t][c][j] = u32x4{base, base + 1, base + 2, base + 3};
              else dst[t][c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + c * UB + j * 32), 0, 0));
        }
      }
    };
    if constexpr (OH) {
      int mytap = -1, myidx = -1;
#pragma unroll
      for (int k = 0; k < K; ++k) { if (idx[k] >= 0) { mytap = k; myidx = idx[k]; } }
      u32x4 a1[UN][NJ];
      const unsigned base1 = (unsigned)myidx * (unsigned)in_ld_b + lane_off;
#pragma unroll
      for (int c = 0; c < UN; ++c)
#pragma unroll
        for (int j = 0; j < NJ; ++j) a1[c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base1 + c * UB + j * 32), 0, 0));
      if constexpr (X3) {
        // fp32 rows, split-bf16 contraction: the one gathered row is split into its hi | lo halves ONCE, then routed to its tap like the
        // 16-bit form (the gather form of the parity-fast mode splits eight gathered rows per output row, seven of them absent)
        u32x4 ah1[UN][2], al1[UN][2];
#pragma unroll
        for (int c = 0; c < UN; ++c)
#pragma unroll
          for (int J = 0; J < 2; ++J) x3_split8(a1[c][2 * J], a1[c][2 * J + 1], ah1[c][J], al1[c][J]);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const bool mine = mytap == k;
#pragma unroll
          for (int c = 0; c < UN; ++c)
#pragma unroll
            for (int J = 0; J < 2; ++J) {
              u32x4 avh, avl;
#pragma unroll
              for (int q = 0; q < 4; ++q) { avh[q] = mine ? ah1[c][J][q] : 0u; avl[q] = mine ? al1[c][J][q] : 0u; }
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) {
                const char* wb = wl + ((k * UN + c) * COUT + nb * 32) * UB;
                const u32x4 bh = *reinterpret_cast<const u32x4*>(wb + (((2 * J + fh) ^ swz) * 16));
                const u32x4 blo = *reinterpret_cast<const u32x4*>(wb + (((4 + 2 * J + fh) ^ swz) * 16));
                mma16_x3(acc[nb], avh, avl, bh, blo);
              }
            }
        }
      } else
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const bool mine = mytap == k;
#pragma unroll
        for (int c = 0; c < UN; ++c)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            u32x4 av;
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = mine ? a1[c][j][q] : 0u;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              const u32x4 bf = *reinterpret_cast<const u32x4*>(wl + ((k * UN + c) * COUT + nb * 32) * UB + (((2 * j + fh) ^ swz) * 16));
              mma16<BF16>(acc[nb], av, bf);
            }
          }
      }
    } else {
    if constexpr (TM) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    tick(0);                                           // 0: rulebook entries (behind the previous tile's stores: vmcnt is in-order)
    issue(0, a[0]);
    tick(1);                                           // 1: issuing the first group of gathers
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) issue(g + 1, a[(g + 1) & 1]);
      if constexpr (PF) {
        if (g == (NG >= 2 ? NG - 2 : 0)) {                      // all gathers of this tile are out: next rulebook words, this residual
          const int lt2 = lt + nblk * WAVES;
          load_words(lt2 < t8 ? tbase + lt2 : ntiles, twn);
        }
      }
      tick(1);
      if constexpr (TM) { if (g + 1 < NG) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * UN * NJ) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      tick(2);                                         // 2: waiting for group g
#pragma unroll
      for (int t = 0; t < G; ++t) {
        const int k = g * G + t;
        if (k < K) {
          if constexpr (X3) {
#pragma unroll
            for (int c = 0; c < UN; ++c)
#pragma unroll
              for (int J = 0; J < 2; ++J) {
                u32x4 ah, al;
                x3_split8(a[g & 1][t][c][2 * J], a[g & 1][t][c][2 * J + 1], ah, al);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                  const char* wb = wl + ((k * UN + c) * COUT + nb * 32) * UB;
                  const u32x4 bh = *reinterpret_cast<const u32x4*>(wb + (((2 * J + fh) ^ swz) * 16));
                  const u32x4 blo = *reinterpret_cast<const u32x4*>(wb + (((4 + 2 * J + fh) ^ swz) * 16));
                  mma16_x3(acc[nb], ah, al, bh, blo);
                }
              }
          } else
#pragma unroll
          for (int c = 0; c < UN; ++c)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) {
                const u32x4 bf = *reinterpret_cast<const u32x4*>(wl + ((k * UN + c) * COUT + nb * 32) * UB + (((2 * j + fh) ^ swz) * 16));
                if constexpr (ABL & 2) acc[nb][(k + j) & 15] += __uint_as_float(a[g & 1][t][c][j][0] ^ bf[1]);
                else mma16<BF16>(acc[nb], a[g & 1][t][c][j], bf);
              }
            }
        }
      }
      tick(3);                                         // 3: LDS weight fragments + MFMAs
    }
    }

    // epilogue (wave-private): acc -> LDS fp32 -> rows as 8-channel vectors
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) ew[((r & 3) + 8 * (r >> 2) + 4 * fh) * EP + nb * 32 + fi] = acc[nb][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (!TR) {
    for (int e = lane; e < 32 * VROW; e += 64) {
      const int rr = e / VROW, cvv = e % VROW;
      const int64_t orow = (int64_t)tile * 32 + rr;
      if (orow >= p.n_out) continue;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8), v1 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8 + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      if constexpr (ABL & 4) { if (v[0] == 1.2345e30f) epi_views8<BF16>(p, orow, cvv * 8, v); }
      else epi_views8<BF16>(p, orow, cvv * 8, v);
    }
    } else {
      // training mode: the row stage writes the summands back into the tile, every lane then adds its NB columns (tl_conv_internal.h)
      constexpr int IT = VROW / 2;                            // 32 * VROW / 64 row vectors per lane
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int e = lane + 64 * it, rr = e / VROW, cvv = e % VROW;
        const int64_t orow = (int64_t)tile * 32 + rr;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8), v1 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]}, q1[8];
        if (orow < p.n_out) epi_views8_red<BF16>(p, orow, cvv * 8, v, q1);
        else {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = 0.f;
        }
        *reinterpret_cast<f32x4*>(ew + rr * EP + cvv * 8) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(ew + rr * EP + cvv * 8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float s0[NB], s1[NB];
      if (p.epi_mode == TL_EPI_STATS) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) tile_colsum<true>(ew, EP, nb * 32 + fi, fh, s0[nb], s1[nb]);
      } else {
        float dummy;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) tile_colsum<false>(ew, EP, nb * 32 + fi, fh, s0[nb], dummy);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < IT; ++it) {                     // g -> g * xhat in place (x re-read: a cache hit)
          const int e = lane + 64 * it, rr = e / VROW, cvv = e % VROW;
          const int64_t orow = (int64_t)tile * 32 + rr;
          if (orow < p.n_out) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8), v1 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8 + 4);
            const float g[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            float q1[8];
            epi_bnb_q1<BF16>(p, orow, cvv * 8, g, q1);
            *reinterpret_cast<f32x4*>(ew + rr * EP + cvv * 8) = f32x4{q1[0], q1[1], q1[2], q1[3]};
            *reinterpret_cast<f32x4*>(ew + rr * EP + cvv * 8 + 4) = f32x4{q1[4], q1[5], q1[6], q1[7]};
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) tile_colsum<false>(ew, EP, nb * 32 + fi, fh, s1[nb], dummy);
      }
      if (lane < 32) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { dw[nb * 32 + fi] += (double)s0[nb]; dw[COUT + nb * 32 + fi] += (double)s1[nb]; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    tick(4);                                           // 4: epilogue (LDS transposition, residual read, stores issued)
    if constexpr (TM) tm[5] += 1;
  }
  if constexpr (TM) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) atomicAdd(&g_tmd[i], tm[i]);
    }
  }
  if constexpr (TR) {                                            // the waves' column sums, added in wave order -> the workgroup's partial row
    __syncthreads();
    for (int e = tid; e < 2 * COUT; e += WAVES * 64) {
      double t = 0.0;
      for (int w = 0; w < WAVES; ++w) t += Ds[w * 2 * COUT + e];
      p.red_part[(int64_t)blockIdx.x * 2 * COUT + e] = t;
    }
  }
}

// ------------------------------------------------------------------ the 4-channel input conv on the matrix cores
// Cin = 4, Cout = 32, K <= 28 (reference tree_learn.py:37-39, `input_conv`).  The reduction index is (tap, channel):
// 27 x 4 = 108 -> seven 32x32x16 MFMA steps.  Lane (i, h) of step s holds channels 0..3 of taps 4s+2h and 4s+2h+1 of
// row i: two 8-byte bounds-checked gathers; the weight fragments (7 x 16 B per lane) live in registers for the whole
// kernel.  HBM-bound on the 108 B/voxel rulebook read and the output writes.
// CT: rulebook in column form (p.ctab), the ten words of the next tile requested while this tile's gathers are in flight.
template <int WAVES, bool CT = false, bool TR = false>
__global__ void __launch_bounds__(WAVES * 64) k_conv_in4(ConvP p, int ntiles) {
  constexpr int EP = 36;
  __shared__ float Es[WAVES][32][EP];
  __shared__ double Ds[TR ? WAVES : 1][2][32];             // training mode (TR): the waves' column sums
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if constexpr (TR) { if (lane < 32) { Ds[wv][0][lane] = 0.0; Ds[wv][1][lane] = 0.0; } }
  const int fi = lane & 31, fh = lane >> 5;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  // B fragments: W packed [K][32][4] bf16 = 8 B per (tap, column)
  u32x4 bfrag[7];
  const u32x2* w2 = reinterpret_cast<const u32x2*>(p.w);
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int t0 = 4 * s + 2 * fh, t1 = t0 + 1;
    const u32x2 z = {0u, 0u};
    const u32x2 b0 = t0 < p.K ? w2[t0 * 32 + fi] : z, b1 = t1 < p.K ? w2[t1 * 32 + fi] : z;
    bfrag[s] = u32x4{b0[0], b0[1], b1[0], b1[1]};
  }
  const int in_ld_b = (int)(p.in_ld * 2);
  const int64_t in_bytes = ((int64_t)p.n_in - 1) * in_ld_b + 8;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)in_bytes, 0x00020000);
  float* ew = &Es[wv][0][0];
  [[maybe_unused]] int twn[10];
  auto load_words = [&](int tile_) __attribute__((always_inline)) {
    const int64_t row = (int64_t)tile_ * 32 + fi;
    const bool rvalid = tile_ < ntiles && row < p.n_out;
#pragma unroll
    for (int k = 0; k < 10; ++k) twn[k] = rvalid ? p.ctab[(int64_t)k * p.n_out + row] : (k < 9 ? -1 : 0);
  };
  if constexpr (CT) load_words(blockIdx.x * WAVES + wv);
  for (int tile = blockIdx.x * WAVES + wv; tile < ntiles; tile += gridDim.x * WAVES) {
    const int64_t row = (int64_t)tile * 32 + fi;
    const bool rvalid = row < p.n_out;
    int idx[14];
    if constexpr (CT) {
      auto tap = [&](int k) __attribute__((always_inline)) {          // k is a compile-time constant after unrolling
        if (k >= 27) return -1;
        const uint32_t m = ((uint32_t)twn[9] >> (3 * (k / 3))) & 7u;
        const int d = k % 3;
        return ((m >> d) & 1u) ? twn[k / 3] + __builtin_popcount(m & ((1u << d) - 1u)) : -1;
      };
#pragma unroll
      for (int q = 0; q < 14; ++q) {
        const int t0 = 4 * (q >> 1) + (q & 1);
        const int a0 = tap(t0), a1 = tap(t0 + 2);
        idx[q] = fh ? a1 : a0;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 14; ++q) {
        const int t = 4 * (q >> 1) + 2 * fh + (q & 1);
        idx[q] = (rvalid && t < p.K) ? p.table[(int64_t)t * p.n_out + row] : -1;
      }
    }
    u32x2 g[14];
#pragma unroll
    for (int q = 0; q < 14; ++q)
      g[q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)((unsigned)idx[q] * (unsigned)in_ld_b), 0, 0));
    if constexpr (CT) load_words(tile + gridDim.x * WAVES);          // after this tile's gathers: no gather wait includes it
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      const u32x4 af = {g[2 * s][0], g[2 * s][1], g[2 * s + 1][0], g[2 * s + 1][1]};
      acc = h16_mfma(af, bfrag[s], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) ew[((r & 3) + 8 * (r >> 2) + 4 * fh) * EP + fi] = acc[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (TR) {
      float r0 = 0.f, r1 = 0.f;
      epi_block32<true, EP>(p, ew, lane, (int64_t)tile * 32, 0, r0, r1);
      if (lane < 32) { Ds[wv][0][lane] += (double)r0; Ds[wv][1][lane] += (double)r1; }
    } else {
#pragma unroll
      for (int e0 = 0; e0 < 2; ++e0) {
        const int e = lane + e0 * 64, rr = e >> 2, cvv = e & 3;
        const int64_t orow = (int64_t)tile * 32 + rr;
        if (orow < p.n_out) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8), v1 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8 + 4);
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          epi_views8<true>(p, orow, cvv * 8, v);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (TR) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64; e += WAVES * 64) {
      double t = 0.0;
      for (int w = 0; w < WAVES; ++w) t += Ds[w][e >> 5][e & 31];
      p.red_part[(int64_t)blockIdx.x * 64 + e] = t;
    }
  }
}

// ------------------------------------------------------------------ the input conv of an all-ones input
// With use_feats = False and use_coords = False (the reference's default flags: tree_learn.py:129-167 `voxelize` then feeds ones) every
// voxel feature is 1, so out[o][co] = sum over the PRESENT taps k of S[k][co], S[k][co] = sum_c W[k][co][c]: no gather at all, just the
// 27-bit presence mask of the column-form rulebook (4 B / voxel) and a 27 x 32 table in LDS.  Write-bound (the output views).
// One thread = one row x 8 channels.
template <bool BF16>
__global__ void __launch_bounds__(256) k_conv_ones27(ConvP p) {
  __shared__ __attribute__((aligned(16))) float S[27][32];
  for (int e = threadIdx.x; e < 27 * 32; e += 256) {                        // weights [27][32][Cin] in the launch's storage type
    float t = 0.f;
    if constexpr (BF16) {
      const uint16_t* w = (const uint16_t*)p.w;
      for (int c = 0; c < p.Cin; ++c) t += bf16_lo((uint32_t)w[(int64_t)e * p.Cin + c]);
    } else {
      const float* w = (const float*)p.w;
      for (int c = 0; c < p.Cin; ++c) t += w[(int64_t)e * p.Cin + c];
    }
    S[e >> 5][e & 31] = t;
  }
  __syncthreads();
  const int32_t* mask = p.blk_pmask ? p.blk_pmask : p.ctab + (int64_t)9 * p.n_out;   // block-local rows: tl_blk_build's presence masks
  const int64_t total = p.n_out * 4;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t row = t >> 2; const int c0 = (int)(t & 3) * 8;
    uint32_t m = (uint32_t)mask[row];
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    while (m) {
      const int k = __builtin_ctz(m); m &= m - 1;
      const f32x4 a = *reinterpret_cast<const f32x4*>(&S[k][c0]), b = *reinterpret_cast<const f32x4*>(&S[k][c0 + 4]);
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[q] += a[q]; v[q + 4] += b[q]; }
    }
    epi_views8<BF16>(p, row, c0, v);
  }
}

int g_direct_walk = 0;

template <bool BF16, int K, int NB, int UN, int G, int WAVES, int ABL = 0, bool CT = false, bool TR = false, bool OH = false, bool X3 = false>
int launch(const ConvP& p_, hipStream_t s) {
  ConvP p = p_;
  if constexpr (X3) {
    if (p.epi_mode != TL_EPI_NONE) return TL_ERR_UNSUPPORTED;
    p.w = p.w_x3;
  }
  if constexpr (!TR && BF16 && ABL == 0) {
    if (p.epi_mode != TL_EPI_NONE) return launch<BF16, K, NB, UN, G, WAVES, ABL, CT, true, OH>(p, s);  // training-mode epilogue: its own instantiation
  } else if constexpr (!TR) {
    if (p.epi_mode != TL_EPI_NONE) return TL_ERR_UNSUPPORTED;
  }
  if constexpr (TR && NB >= 3) return TL_ERR_UNSUPPORTED;       // 96 output channels: six row vectors per lane spill; the stream kernels take these
  else {
  constexpr int UB = BF16 ? 64 : 128;
  const size_t lds = (size_t)K * UN * NB * 32 * UB + (size_t)WAVES * 32 * (NB * 32 + 4) * 4 + (TR ? (size_t)WAVES * 2 * NB * 32 * 8 : 0);
  if (lds > 160 * 1024) return TL_ERR_UNSUPPORTED;
  static TlAttrOnce attr_once;                     // per kernel instantiation AND device (the attribute is per device)
  if (!tl_lds_attr(attr_once, reinterpret_cast<const void*>(&k_conv_direct<BF16, K, NB, UN, G, WAVES, ABL, CT, TR, OH, X3>), 160 * 1024)) return TL_ERR_LAUNCH;
  const int ntiles = (int)tl_cdiv(p.n_out, 32);
  const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  int grid = 256 * (per_cu > 2 ? 2 : per_cu);
  const int need = (int)tl_cdiv(ntiles, WAVES);
  if (grid > need) grid = need;
  k_conv_direct<BF16, K, NB, UN, G, WAVES, ABL, CT, TR, OH, X3><<<grid, WAVES * 64, lds, s>>>(p, ntiles, (g_direct_walk && grid % 8 == 0) ? 1 : 0);
  if (p.red_nparts) *p.red_nparts = grid;
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
  }
}

int g_direct_abl = 0;

template <bool BF16, int K, int G>
int dispatch(const ConvP& p, hipStream_t s) {
  const int nb = p.Cout / 32, un = p.Cin / 32;
#ifdef TL_DEV
  if constexpr (BF16 && K == 27) {
    if (g_direct_abl && g_direct_abl < 14 && nb == 1 && un == 1) {
      switch (g_direct_abl) {
        case 1: return launch<true, 27, 1, 1, 3, 16, 1>(p, s);
        case 2: return launch<true, 27, 1, 1, 3, 16, 2>(p, s);
        case 3: return launch<true, 27, 1, 1, 3, 16, 4>(p, s);
        case 4: return launch<true, 27, 1, 1, 3, 16, 8>(p, s);
        case 5: return launch<true, 27, 1, 1, 3, 16, 9>(p, s);
        case 6: return launch<true, 27, 1, 1, 3, 16, 5>(p, s);
        case 7: return launch<true, 27, 1, 1, 3, 8>(p, s);
        case 8: return launch<true, 27, 1, 1, 9, 16>(p, s);
        case 9: return launch<true, 27, 1, 1, 1, 16>(p, s);
        case 10: return launch<true, 27, 1, 1, 3, 16, 16>(p, s);     // segment timers
      }
    }
  }
#endif
  const size_t wbytes = (size_t)K * p.Cout * p.Cin * (BF16 ? 2 : 4);
  if constexpr (BF16 && K == 27) {
    // column-form rulebook (level 1), its words requested one tile ahead: 32 -> 32 0.200 -> 0.195 ms on top of the 0.243 -> 0.226 of
    // the 40-B form itself; 64 -> 32 0.51 -> 0.33 ms (without the look-ahead the decode in front of its gathers had made it slower)
    if (p.ctab && (g_direct_abl == 0 || g_direct_abl >= 14) && nb == 1 && un == 1) return launch<true, 27, 1, 1, G, 16, 0, true>(p, s);
    if (p.ctab && g_direct_abl != 14 && nb == 1 && un == 2) return launch<true, 27, 1, 2, G, 8, 0, true>(p, s);
#ifdef TL_DEV
    if (p.ctab && g_direct_abl == 13 && nb == 1 && un == 1) return launch<true, 27, 1, 1, G, 16, 32, true>(p, s);   // rulebook words not requested ahead
#endif
  }
  if constexpr (BF16 && K == 8) {
    // inverse conv of level 1 (64 -> 32, one valid entry per row): the row gathered once, no per-tap barrier (the stream kernel's one-hot
    // form took 0.171 ms for 0.38 GB of traffic: eight barrier steps per 256 rows)
    if (p.one_hot && nb == 1 && un == 2) return launch<true, 8, 1, 2, G, 16, 0, false, false, true>(p, s);
  }
  if constexpr (BF16 && K == 1) {
    // the 1x1 convs of the decoder blocks (2C -> C on the skip concat) of levels 2-4: pure streaming, weights resident (level 2: 0.108 ms
    // on the tile kernel for 424 MB = 3.9 TB/s; the level-1 shape runs at 5.1 TB/s here)
    if (nb == 2 && un == 4) return launch<true, 1, 2, 4, 1, 16>(p, s);
    if (nb == 3 && un == 6) return launch<true, 1, 3, 6, 1, 8>(p, s);
    if (nb == 4 && un == 8) return launch<true, 1, 4, 8, 1, 8>(p, s);
  }
  if constexpr (!BF16 && K == 1) {
    // the 2C -> C 1x1 conv of the level-2 decoder block on fp32 rows (exact or split-bf16): pure streaming with resident weights, as in bf16
    // (the stream kernel moved its 0.86 GB at 2.3 TB/s)
    if (nb == 2 && un == 4 && p.epi_mode == TL_EPI_NONE) {
      if (p.w_x3) return launch<false, 1, 2, 4, 1, 8, 0, false, false, false, true>(p, s);
      return launch<false, 1, 2, 4, 1, 8>(p, s);
    }
  }
  if constexpr (!BF16 && K == 8) {
    // ... and the same conv in the parity-fast mode (fp32 rows, split-bf16 weights: 64 KB resident + twelve epilogue buffers)
    if (p.one_hot && p.w_x3 && p.epi_mode == TL_EPI_NONE && nb == 1 && un == 2) return launch<false, 8, 1, 2, G, 12, 0, false, false, true, true>(p, s);
  }
  // 16-wave workgroups (bf16 only: 128 VGPRs suffice) when the weights leave room for 16 epilogue buffers
#define TL_D(NB_, UN_)                                                                                               \
  if (nb == NB_ && un == UN_) {                                                                                      \
    if constexpr (BF16) return wbytes <= 64 * 1024 ? launch<true, K, NB_, UN_, G, 16>(p, s) : launch<true, K, NB_, UN_, G, 8>(p, s); \
    else {                                                                                                           \
      if (p.w_x3 && p.epi_mode == TL_EPI_NONE)       /* split-bf16: latency-bound, so as many waves as the LDS holds beside 110 KB of weights */ \
        return launch<false, K, NB_, UN_, (G > 2 ? 2 : G), (K == 27 && NB_ == 1 && UN_ == 1 ? 11 : 8), 0, false, false, false, true>(p, s);         \
      return launch<false, K, NB_, UN_, (G > 2 ? 2 : G), 8>(p, s);                                                   \
    }                                                                                                                \
  }
  TL_D(1, 1) TL_D(1, 2) TL_D(2, 1) TL_D(2, 2) TL_D(2, 3) TL_D(3, 2) TL_D(1, 3) TL_D(3, 1)
#undef TL_D
  return TL_ERR_UNSUPPORTED;
}

}  // namespace

// all-ones input (tl_conv_args.in_all_ones): bf16, K = 27 with the column-form rulebook, Cout = 32
int tl_launch_conv_ones27(const ConvP& p, int dtype, hipStream_t s) {
  if (p.K != 27 || p.Cout != 32 || (!p.ctab && !p.blk_pmask) || p.Cin <= 0 || p.Cin > 64 || p.in_scale || p.in_relu || p.epi_mode != TL_EPI_NONE) return TL_ERR_UNSUPPORTED;
  if (dtype == TL_BF16) k_conv_ones27<true><<<tl_grid(p.n_out * 4, 256), 256, 0, s>>>(p);
  else k_conv_ones27<false><<<tl_grid(p.n_out * 4, 256), 256, 0, s>>>(p);          // fp32 storage (the exact and the bf16x3 parity modes)
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}


#ifdef TL_DEV
// developer hook (dev build only), not part of the C ABI: mode 0..99 = ablation variant of the 32->32 kernel, 1000 / 1001 = tile walk of every direct launch
extern "C" int tl_dev_direct_abl(int mode) { if (mode >= 1000) g_direct_walk = mode - 1000; else g_direct_abl = mode; return TL_OK; }
extern "C" int tl_dev_direct_tm(unsigned long long* out8) {          // read and clear the segment timers of ablation mode 10
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tmd), sizeof(g_tmd)) != hipSuccess) return TL_ERR_LAUNCH;
  unsigned long long z[8] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tmd), z, sizeof(z)) == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}
#endif

// Eligibility (beyond the tile kernel's alignment rules): no gather-side prologue, whole weight tensor + epilogue
// scratch within LDS, input view below 4 GB.  Returns TL_ERR_UNSUPPORTED when the shape is not covered.
int tl_launch_conv_direct(const ConvP& p, int dtype, hipStream_t s) {
  if (p.in_scale || p.in_relu) return TL_ERR_UNSUPPORTED;
  if (p.one_hot == 2 && !(p.K == 8 && p.Cin == 64 && p.Cout == 32 && p.epi_mode == TL_EPI_NONE && (dtype == TL_BF16 || p.w_x3))) return TL_ERR_UNSUPPORTED;
  const int eb = dtype == TL_BF16 ? 2 : 4;
  if (dtype == TL_BF16 && p.Cin == 4 && p.Cout == 32 && p.K <= 28 && p.table && p.in_ld % 4 == 0 && (int64_t)p.n_in * p.in_ld * 2 < 0xFFFF0000ll) {
    const int ntiles = (int)tl_cdiv(p.n_out, 32);
    int grid = (int)tl_cdiv(ntiles, 8);
    if (grid > 2048) grid = 2048;
    if (p.epi_mode != TL_EPI_NONE) {
      if (p.ctab && p.K == 27) k_conv_in4<8, true, true><<<grid, 512, 0, s>>>(p, ntiles);
      else k_conv_in4<8, false, true><<<grid, 512, 0, s>>>(p, ntiles);
    } else if (p.ctab && p.K == 27 && g_direct_abl != 15) k_conv_in4<8, true><<<grid, 512, 0, s>>>(p, ntiles);
    else k_conv_in4<8><<<grid, 512, 0, s>>>(p, ntiles);
    if (p.red_nparts) *p.red_nparts = grid;
    return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
  }
  if (p.Cin % 32 || p.Cout % 32) return TL_ERR_UNSUPPORTED;
  const int64_t ld_b = p.in_ld * eb, in_bytes = (p.n_in - 1) * ld_b + (int64_t)p.Cin * eb;
  if (!(in_bytes > 0 && in_bytes + 2 * ld_b < 0xFFFFFFFFll)) return TL_ERR_UNSUPPORTED;
  if ((size_t)p.K * p.Cout * p.Cin * eb > 112 * 1024) return TL_ERR_UNSUPPORTED;
  if (dtype == TL_BF16) {
    switch (p.K) {
      case 27: return dispatch<true, 27, 3>(p, s);
      case 8: return dispatch<true, 8, 2>(p, s);
      case 1: return dispatch<true, 1, 1>(p, s);
    }
  } else {
    switch (p.K) {
      case 27: return dispatch<false, 27, 3>(p, s);
      case 8: return dispatch<false, 8, 2>(p, s);
      case 1: return dispatch<false, 1, 1>(p, s);
    }
  }
  return TL_ERR_UNSUPPORTED;
}
