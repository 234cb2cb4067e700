// Sparse conv, "stream" form (bf16 and fp32): mid levels (C = 64..128) whose weights do not fit in LDS as a whole.
//
// Same idea as tl_conv_direct.hip -- every wave owns a 32-row output tile and gathers its MFMA A-fragments
// straight from global memory into registers with bounds-checked buffer loads (absent neighbour -> zeros), prefetched
// DA taps ahead -- but the weights are streamed: the workgroup (8 waves = 256 rows) stages ONE tap's [Cout x Cin]
// slice per step in a double-buffered LDS tile (16 B per thread), so the per-step barrier only guards 8-32 KB
// of weights; the A operand never touches LDS.  Compared with the tile kernel this removes two thirds of the LDS
// traffic, all staging VALU work and the rulebook-in-LDS phase.  The tap loop is fully unrolled and branch-free
// (counted vmcnt survives), all K taps are contracted.  Deterministic.
#include "tl_conv_internal.h"

namespace {

constexpr int WAVES = 8;
constexpr int NT = WAVES * 64;

template <bool BF16, int K, int NB, int UN, int DA, int OCC>
__global__ void __launch_bounds__(NT, OCC) k_conv_stream(ConvP p) {
  constexpr int EB = BF16 ? 2 : 4, UB = 32 * EB, NJ = UB / 32, SLOTS = UB / 16;
  constexpr int COUT = NB * 32, CIN = UN * 32;
  constexpr int BROW = CIN * EB + 16;                  // LDS pitch of a weight row (one output channel, one tap): +16 B pad =>
                                                      // ds_read_b128 of 16 different rows at one column is conflict-free
  constexpr int BSLOTS = UN * SLOTS;                  // 16-B vectors per weight row
  constexpr int BVEC = COUT * BSLOTS;                 // 16-B vectors per tap
  constexpr int BPT = (BVEC + NT - 1) / NT;           // vectors per thread per tap
  constexpr int EP = 32 + 4;                          // epilogue pitch (floats), one 32-column block at a time
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Bs = smem;                                                          // [2][COUT][BROW]
  float* Es = reinterpret_cast<float*>(smem);                               // epilogue alias of Bs: [WAVES][32][EP]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  const int tile = xcd_tile(blockIdx.x, p.nblk);
  const int64_t r0 = (int64_t)tile * (WAVES * 32) + wv * 32;
  const int64_t row = r0 + fi;
  const bool rvalid = row < p.n_out;

  int idx[K];
#pragma unroll
  for (int k = 0; k < K; ++k) idx[k] = rvalid ? (p.table ? p.table[(int64_t)k * p.n_out + row] : (int)row) : -1;

  const int in_ld_b = (int)(p.in_ld * EB);
  const int64_t in_bytes = ((int64_t)p.n_in - 1) * in_ld_b + (int64_t)CIN * EB;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)in_bytes, 0x00020000);
  const unsigned lane_off = (unsigned)(fh * 16);
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

  u32x4 a[DA][UN][NJ];
  u32x4 bw[BPT];
  auto issue_a = [&](int k, u32x4 (&dst)[UN][NJ]) __attribute__((always_inline)) {
    const unsigned base = (unsigned)idx[k] * (unsigned)in_ld_b + lane_off;
#pragma unroll
    for (int c = 0; c < UN; ++c)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        dst[c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + c * UB + j * 32), 0, 0));
  };
  auto load_b = [&](int k) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const int v = tid + q * NT;
      bw[q] = wsrc[(int64_t)k * BVEC + (BVEC % NT == 0 ? v : min(v, BVEC - 1))];
    }
  };
  auto store_b = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const int v = tid + q * NT;
      const int n = v / BSLOTS, s = v % BSLOTS;
      if (BVEC % NT == 0 || v < BVEC) *reinterpret_cast<u32x4*>(Bs + buf * COUT * BROW + n * BROW + s * 16) = bw[q];
    }
  };

  // prologue: weights of tap 0 -> LDS, taps 0..DA-1 of A in flight, weights of tap 1 in registers
  load_b(0);
#pragma unroll
  for (int d = 0; d < DA; ++d) if (d < K) issue_a(d, a[d]);
  store_b(0);
  if (K > 1) load_b(1);
  __syncthreads();

#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k + 1 < K) store_b((k + 1) & 1);                  // tap k+1's weights (loaded last step) -> other buffer
    if (k + 2 < K) load_b(k + 2);
    const char* bl = Bs + (k & 1) * COUT * BROW + fi * BROW;
#pragma unroll
    for (int c = 0; c < UN; ++c)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int slot = c * SLOTS + 2 * j + fh;
          const u32x4 bf = *reinterpret_cast<const u32x4*>(bl + nb * 32 * BROW + slot * 16);
          mma16<BF16>(acc[nb], a[k % DA][c][j], bf);
        }
      }
    if (k + DA < K) issue_a(k + DA, a[k % DA]);
    if (k + 1 < K) __syncthreads();
  }

  // epilogue, one 32-column block at a time through a wave-private LDS transposition buffer (aliases the weight tiles)
  __syncthreads();
  float* ew = Es + wv * 32 * EP;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) ew[((r & 3) + 8 * (r >> 2) + 4 * fh) * EP + fi] = acc[nb][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e0 = 0; e0 < 2; ++e0) {
      const int e = lane + e0 * 64;                        // 32 rows x 4 vectors
      const int rr = e >> 2, cvv = e & 3;
      const int64_t orow = r0 + rr;
      if (orow < p.n_out) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8), v1 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        epi_views8<BF16>(p, orow, nb * 32 + cvv * 8, v);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <bool BF16, int K, int NB, int UN, int DA>
int launch(ConvP p, hipStream_t s) {
  constexpr int EB = BF16 ? 2 : 4;
  constexpr int OCC = (NB * 16 + DA * UN * (BF16 ? 8 : 16) + 40 <= 118) ? 4 : 2;     // rough VGPR need -> waves per SIMD to ask for
  const size_t wt = 2 * (size_t)NB * 32 * (UN * 32 * EB + 16), ep = (size_t)WAVES * 32 * 36 * 4;
  const size_t lds = wt > ep ? wt : ep;
  if (lds > 160 * 1024) return TL_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_stream<BF16, K, NB, UN, DA, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return TL_ERR_LAUNCH;
    attr_set = true;
  }
  p.nblk = (int)tl_cdiv(p.n_out, WAVES * 32);
  k_conv_stream<BF16, K, NB, UN, DA, OCC><<<p.nblk, NT, lds, s>>>(p);
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

template <bool BF16, int K>
int dispatch(const ConvP& p, hipStream_t s) {
  const int nb = p.Cout / 32, un = p.Cin / 32;
  // (NB, UN, prefetch depth bf16, prefetch depth fp32)
#define TL_S(NB_, UN_, DB_, DF_) if (nb == NB_ && un == UN_) return launch<BF16, K, NB_, UN_, (BF16 ? DB_ : DF_)>(p, s);
  TL_S(2, 2, 3, 2) TL_S(2, 4, 2, 1) TL_S(3, 3, 2, 1) TL_S(3, 6, 1, 1) TL_S(4, 4, 2, 1) TL_S(2, 3, 3, 1) TL_S(3, 2, 3, 2) TL_S(3, 4, 2, 1)
  TL_S(4, 3, 2, 1) TL_S(1, 2, 3, 2) TL_S(2, 1, 3, 2) TL_S(1, 1, 3, 2)
#undef TL_S
  return TL_ERR_UNSUPPORTED;
}

}  // namespace

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
