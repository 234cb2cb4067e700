// Sparse conv, "stream" form (bf16 and fp32): mid levels (C = 64..128) whose weights do not fit in LDS as a whole.
//
// Same idea as tl_conv_direct.hip -- every wave owns its output rows end to end and gathers its MFMA A-fragments
// straight from global memory into registers with bounds-checked buffer loads (absent neighbour -> zeros), prefetched
// DA taps ahead -- but the weights are streamed: the workgroup (8 waves) stages ONE tap's [Cout x Cin] slice per step
// in a double-buffered LDS tile (16 B per thread), so the per-step barrier only guards 8-32 KB of weights; the A
// operand never touches LDS.  The tap loop is fully unrolled and branch-free (counted vmcnt survives), all K taps are
// contracted.  Deterministic.
//
// RB = 32-row blocks per wave.  Ablations on the level-2 conv (C = 64, bf16): without gathers 0.28 of 0.43 ms remain,
// without the barrier nothing changes, deeper prefetch changes nothing -- what remains is LDS read bandwidth: with one
// row block per wave every 32x32x16 MFMA (32 cycles on one SIMD) needs its own 1 KB ds_read_b128 weight fragment, i.e.
// 4 SIMDs x 1 KB / 32 clk = 128 B/clk = the whole LDS read rate of a CU.  With RB = 2 each weight fragment feeds two
// MFMAs (two row blocks), halving the LDS traffic per flop.
#include "tl_conv_internal.h"
#include <atomic>

namespace {

constexpr int WAVES = 8;
constexpr int NT = WAVES * 64;

__device__ unsigned long long g_tm[8];   // developer timing mode (tl_dev_stream_tm): cycles summed over waves per step segment

// OH (inverse conv, one valid table entry per output row): the row's single input row is gathered ONCE and routed to its
// tap by a per-lane select, instead of K gathers of which K - 1 are out of range.
// X3 (fp32 storage only): the contraction runs as split-bf16 products (tl_conv_internal.h: mma16_x3) on weights in the tl_pack_weight_x3 form
template <bool BF16, int K, int NB, int UN, int DA, int RB, int OCC, bool TM = false, bool OH = false, bool X3 = false>
__global__ void __launch_bounds__(NT, OCC) k_conv_stream(ConvP p) {
  static_assert(!X3 || !BF16, "the split-bf16 contraction reads fp32 rows");
  constexpr int EB = BF16 ? 2 : 4, UB = 32 * EB, NJ = UB / 32, SLOTS = UB / 16;
  constexpr int COUT = NB * 32, CIN = UN * 32;
  constexpr int BROW = CIN * EB + 16;                  // LDS pitch of a weight row (one output channel, one tap): +16 B pad =>
                                                      // ds_read_b128 of 16 different rows at one column is conflict-free
  constexpr int BSLOTS = UN * SLOTS;                  // 16-B vectors per weight row
  constexpr int BVEC = COUT * BSLOTS;                 // 16-B vectors per tap
  constexpr int BPT = (BVEC + NT - 1) / NT;           // vectors per thread per tap
  constexpr int EP = 32 + 4;                          // epilogue pitch (floats), one 32x32 block at a time
  constexpr int WROWS = 32 * RB;                      // rows per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Bs = smem;                                                          // [2][COUT][BROW]
  float* Es = reinterpret_cast<float*>(smem);                               // epilogue alias of Bs: [WAVES][32][EP]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  const int tile = xcd_tile(blockIdx.x, p.nblk);
  const int64_t r0 = (int64_t)tile * (WAVES * WROWS) + wv * WROWS;

  int idx[K][RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int64_t row = r0 + rb * 32 + fi;
    const bool rvalid = row < p.n_out;
    if (K == 27 && p.ctab) {                                 // column form of the rulebook: 40 B per row instead of 108
      int t27[27];
      decode_ctab(p.ctab, p.n_out, row, rvalid, t27);
#pragma unroll
      for (int k = 0; k < K; ++k) idx[k][rb] = t27[k < 27 ? k : 0];
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) idx[k][rb] = rvalid ? (p.table ? p.table[(int64_t)k * p.n_out + row] : (int)row) : -1;
    }
  }

  const int in_ld_b = (int)(p.in_ld * EB);
  const int64_t in_bytes = ((int64_t)p.n_in - 1) * in_ld_b + (int64_t)CIN * EB;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)in_bytes, 0x00020000);
  const unsigned lane_off = (unsigned)(fh * 16);
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);

  f32x16 acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][nb][i] = 0.f;

  // Weights of tap t are requested at step t - WA, *before* that step's gathers, and written to LDS at step t - 1.  With
  // the in-order vmcnt this is the only placement where waiting for them does not also wait for younger gathers: the
  // loads older than weights(t+1) at the top of step t are exactly gathers(<= t), which step t needs anyway.
  constexpr int WA = DA > 2 ? DA : 2, RW = WA - 1;
  u32x4 a[DA][RB][UN][NJ];
  u32x4 bw[RW][BPT], bw0[BPT];
  auto issue_a = [&](int k, u32x4 (&dst)[RB][UN][NJ]) __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const unsigned base = (unsigned)idx[k][rb] * (unsigned)in_ld_b + lane_off;
#pragma unroll
      for (int c = 0; c < UN; ++c)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          dst[rb][c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + c * UB + j * 32), 0, 0));
    }
  };
  auto load_b = [&](int k, u32x4 (&dst)[BPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const int v = tid + q * NT;
      dst[q] = wsrc[(int64_t)k * BVEC + (BVEC % NT == 0 ? v : min(v, BVEC - 1))];
    }
  };
  auto store_b = [&](int buf, const u32x4 (&src)[BPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const int v = tid + q * NT;
      const int n = v / BSLOTS, s = v % BSLOTS;
      if (BVEC % NT == 0 || v < BVEC) *reinterpret_cast<u32x4*>(Bs + buf * COUT * BROW + n * BROW + s * 16) = src[q];
    }
  };

  // prologue: weights of tap 0 -> LDS; taps 1..WA-1 of the weights and 0..DA-1 of A in flight, interleaved in tap order
  load_b(0, bw0);
  [[maybe_unused]] u32x4 a1[RB][UN][NJ];                   // OH: the one gathered row per output row
  [[maybe_unused]] u32x4 ah1[RB][UN][2], al1[RB][UN][2];   // OH && X3: its hi / lo halves, split ONCE (at the first tap, when the row has landed)
  if constexpr (OH) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      int ip = -1;
#pragma unroll
      for (int k = 0; k < K; ++k) ip = max(ip, idx[k][rb]);
      const unsigned base = (unsigned)ip * (unsigned)in_ld_b + lane_off;
#pragma unroll
      for (int c = 0; c < UN; ++c)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          a1[rb][c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + c * UB + j * 32), 0, 0));
    }
  }
#pragma unroll
  for (int d = 0; d < WA; ++d) {
    if (d >= 1 && d < K) load_b(d, bw[d % RW]);
    if constexpr (!OH) { if (d < DA && d < K) issue_a(d, a[d]); }
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
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k + 1 < K) store_b((k + 1) & 1, bw[(k + 1) % RW]);     // tap k+1's weights -> the buffer step k-1 was reading
    if (k + WA < K) load_b(k + WA, bw[(k + WA) % RW]);          // the slot just emptied
    __builtin_amdgcn_sched_barrier(0);                         // keep the request up here (hipcc sinks it below the MFMAs)
    if constexpr (TM) { if (k + DA < K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DA - 1) * (RB * UN * NJ + BPT) + BPT)); else asm volatile("s_waitcnt vmcnt(0)"); }
    tick(0);
    const char* bl = Bs + (k & 1) * COUT * BROW + fi * BROW;
    if constexpr (X3) {
#pragma unroll
      for (int c = 0; c < UN; ++c)
#pragma unroll
        for (int J = 0; J < 2; ++J) {
          u32x4 ah[RB], al[RB];
          if constexpr (OH) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
              if (k == 0) x3_split8(a1[rb][c][2 * J], a1[rb][c][2 * J + 1], ah1[rb][c][J], al1[rb][c][J]);
              const bool mine = idx[k][rb] >= 0;
#pragma unroll
              for (int q = 0; q < 4; ++q) { ah[rb][q] = mine ? ah1[rb][c][J][q] : 0u; al[rb][q] = mine ? al1[rb][c][J][q] : 0u; }
            }
          } else {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) x3_split8(a[k % DA][rb][c][2 * J], a[k % DA][rb][c][2 * J + 1], ah[rb], al[rb]);
          }
          // the three products of a column block form a dependent chain on its accumulator: issue term by term ACROSS the column blocks
          u32x4 bh[NB], blo[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const char* bp = bl + nb * 32 * BROW + (c * SLOTS + 2 * J + fh) * 16;
            bh[nb] = *reinterpret_cast<const u32x4*>(bp); blo[nb] = *reinterpret_cast<const u32x4*>(bp + 64);
          }
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[rb]), __builtin_bit_cast(bf16x8, bh[nb]), acc[rb][nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[rb]), __builtin_bit_cast(bf16x8, blo[nb]), acc[rb][nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[rb]), __builtin_bit_cast(bf16x8, bh[nb]), acc[rb][nb], 0, 0, 0);
          }
        }
    } else
#pragma unroll
    for (int c = 0; c < UN; ++c)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int slot = c * SLOTS + 2 * j + fh;
          const u32x4 bf = *reinterpret_cast<const u32x4*>(bl + nb * 32 * BROW + slot * 16);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {                                                     // one weight fragment, RB MFMAs
            if constexpr (OH) {
              const u32x4 z = {0u, 0u, 0u, 0u};
              mma16<BF16>(acc[rb][nb], idx[k][rb] >= 0 ? a1[rb][c][j] : z, bf);
            } else mma16<BF16>(acc[rb][nb], a[k % DA][rb][c][j], bf);
          }
        }
      }
    tick(1);
    if constexpr (!OH) { if (k + DA < K) issue_a(k + DA, a[k % DA]); }
    tick(2);
    if (k + 1 < K) __syncthreads();
    tick(3);
  }
  if constexpr (TM) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) atomicAdd(&g_tm[i], tm[i]);
      atomicAdd(&g_tm[4], 1ull);
    }
  }

  // epilogue, one 32x32 block at a time through a wave-private LDS transposition buffer (aliases the weight tiles)
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
      for (int r = 0; r < 16; ++r) ew[((r & 3) + 8 * (r >> 2) + 4 * fh) * EP + fi] = acc[rb][nb][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      epi_block32<BF16, EP>(p, ew, lane, r0 + rb * 32, nb * 32, red0[nb], red1[nb]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  if (p.epi_mode != TL_EPI_NONE) epi_finish_wg<WAVES, EP, NB>(p, Es, tid, red0, red1);
}

template <bool BF16, int K, int NB, int UN, int DA, int RB, bool TM = false, bool OH = false, bool X3 = false>
int launch(ConvP p, hipStream_t s) {
  if constexpr (X3) p.w = p.w_x3;
  constexpr int EB = BF16 ? 2 : 4;
  constexpr int BPTL = (NB * 32 * UN * (BF16 ? 4 : 8) + NT - 1) / NT;
  constexpr int VG = RB * NB * 16 + DA * RB * UN * (BF16 ? 8 : 16) + (K <= 8 ? 8 : 27) * RB + 4 * BPTL * (DA > 2 ? DA - 1 : 1) + 20;   // rough VGPR need
  constexpr int OCC = (OH && X3) ? 2 : ((VG <= 120 || (X3 && NB * UN <= 4)) ? 4 : 2);      // (the split-bf16 form is latency-bound at two waves per SIMD: measured;
                                                                                             //  its one-hot form also holds the split row: 48 registers more)
  const size_t wt = 2 * (size_t)NB * 32 * (UN * 32 * EB + 16), ep = (size_t)WAVES * 32 * 36 * 4;
  const size_t lds = wt > ep ? wt : ep;
  if (lds > 160 * 1024) return TL_ERR_UNSUPPORTED;
  static TlAttrOnce attr_once;                     // per kernel instantiation AND device (the attribute is per device)
  if (!tl_lds_attr(attr_once, reinterpret_cast<const void*>(&k_conv_stream<BF16, K, NB, UN, DA, RB, OCC, TM, OH, X3>), 160 * 1024)) return TL_ERR_LAUNCH;
  p.nblk = (int)tl_cdiv(p.n_out, WAVES * 32 * RB);
  k_conv_stream<BF16, K, NB, UN, DA, RB, OCC, TM, OH, X3><<<p.nblk, NT, lds, s>>>(p);
  if (p.red_nparts) *p.red_nparts = p.nblk;
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

int g_stream_da = 0;
int g_stream_tm = 0;     // developer timing mode on the 64->64 bf16 shape
int g_stream_rb = 0;     // bf16: 32-row blocks per wave (tl_set_tuning "stream_rb"): 1, 2, or 0 = measured best per shape (2 for 96->96)

template <bool BF16, int K>
int dispatch(const ConvP& p, hipStream_t s) {
  const int nb = p.Cout / 32, un = p.Cin / 32;
  if constexpr (BF16 && K == 8) {
    if (p.one_hot) {                                          // the up (inverse) convs of the large levels
      if (nb == 2 && un == 3) return launch<true, 8, 2, 3, 1, 1, false, true>(p, s);
      if (nb == 1 && un == 2) return launch<true, 8, 1, 2, 1, 1, false, true>(p, s);
      if (nb == 3 && un == 4) return launch<true, 8, 3, 4, 1, 1, false, true>(p, s);
    }
  }
  if constexpr (!BF16 && K == 8) {
    // ... and the same convs in the parity-fast mode (fp32 rows, split-bf16 weights): the one row gathered and split once instead of eight
    // gathered rows -- seven of them absent -- split per tap (level 2 <- 3: 0.37 ms for 0.39 GB)
    if (p.one_hot == 1 && p.w_x3 && p.epi_mode == TL_EPI_NONE) {
      if (nb == 2 && un == 3) return launch<false, 8, 2, 3, 1, 1, false, true, true>(p, s);
      if (nb == 3 && un == 4) return launch<false, 8, 3, 4, 1, 1, false, true, true>(p, s);
    }
  }
  if constexpr (BF16 && K == 8) {
    // the level-1 -> 2 strided conv (two output views, 1.7 of 8 taps present): two row blocks per wave -- half as many workgroups, each
    // amortising its start (table, first gathers) and drain (the stores of two views) over 512 rows: 0.162 -> 0.124 ms with two views,
    // 0.090 -> 0.086 with one; prefetch depth 4 / 6 / 8: no change; three or four row blocks: 0.128 (tools/dev_k8.py)
    if (!p.one_hot && nb == 2 && un == 1 && g_stream_rb != 1) return launch<true, 8, 2, 1, 3, 2>(p, s);
  }
  if constexpr (BF16 && K == 27) {
    if (g_stream_tm && nb == 2 && un == 2) return launch<true, 27, 2, 2, 3, 1, true>(p, s);
    if (g_stream_da && nb == 2 && un == 2) {                  // developer A/B of the prefetch depth on the 64->64 shape
      if (g_stream_da == 2) return launch<true, 27, 2, 2, 2, 1>(p, s);
      if (g_stream_da == 4) return launch<true, 27, 2, 2, 4, 1>(p, s);
      if (g_stream_da == 5) return launch<true, 27, 2, 2, 5, 1>(p, s);
    }
  }
  // (NB, UN): bf16 with one row block per wave (prefetch depth DB1) or two (DB2, 0 = not offered); fp32: one (depth DF)
#define TL_S(NB_, UN_, DB1_, DB2_, DF_)                                                              \
  if (nb == NB_ && un == UN_) {                                                                      \
    if constexpr (BF16) {                                                                            \
      if (DB2_ > 0 && (g_stream_rb == 2 || (g_stream_rb == 0 && NB_ == 3 && UN_ == 3))) return launch<true, K, NB_, UN_, (DB2_ > 0 ? DB2_ : 1), 2>(p, s); \
      return launch<true, K, NB_, UN_, DB1_, 1>(p, s);                                               \
    } else {                                                                                         \
      if (p.w_x3) return launch<false, K, NB_, UN_, 1, 1, false, false, true>(p, s);   /* split-bf16 contraction */ \
      return launch<false, K, NB_, UN_, DF_, 1>(p, s);                                               \
    }                                                                                                \
  }
  TL_S(2, 2, 3, 2, 2) TL_S(2, 4, 2, 1, 1) TL_S(3, 3, 2, 1, 1) TL_S(3, 6, 1, 0, 1) TL_S(4, 4, 2, 0, 1) TL_S(2, 3, 3, 1, 1) TL_S(3, 2, 3, 2, 2)
  TL_S(3, 4, 2, 1, 1) TL_S(4, 3, 2, 1, 1) TL_S(1, 2, 3, 2, 2) TL_S(2, 1, 3, 2, 2) TL_S(1, 1, 3, 2, 2)
#undef TL_S
  return TL_ERR_UNSUPPORTED;
}

}  // namespace

int tl_stream_set_rb(int rb) { if (rb >= 100) g_stream_da = rb - 100; else g_stream_rb = rb; return TL_OK; }

// Developer hook (not part of the C ABI): switch the per-segment cycle counters on/off, read and clear them.
#ifdef TL_DEV
extern "C" int tl_dev_stream_tm(int enable, unsigned long long* out8) {
  g_stream_tm = enable;
  if (out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tm), sizeof(g_tm)) != hipSuccess) return TL_ERR_LAUNCH;
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tm), z, sizeof(z)) != hipSuccess) return TL_ERR_LAUNCH;
  }
  return TL_OK;
}
#endif

int tl_launch_conv_stream(const ConvP& p, int dtype, hipStream_t s) {
  if (p.in_scale || p.in_relu) return TL_ERR_UNSUPPORTED;
  const int eb = dtype == TL_BF16 ? 2 : 4;
  const int64_t ld_b = p.in_ld * eb, in_bytes = (p.n_in - 1) * ld_b + (int64_t)p.Cin * eb;
  if (!(in_bytes > 0 && in_bytes + 2 * ld_b < 0xFFFFFFFFll)) return TL_ERR_UNSUPPORTED;
  if (dtype == TL_BF16) {
    switch (p.K) {
      case 27: return dispatch<true, 27>(p, s);
      case 8: return dispatch<true, 8>(p, s);
    }
  } else {
    switch (p.K) {
      case 27: return dispatch<false, 27>(p, s);
      case 8: return dispatch<false, 8>(p, s);
    }
  }
  return TL_ERR_UNSUPPORTED;
}
