// bf16 sparse conv, "stream" form: mid levels (C = 64..128) whose weights do not fit in LDS as a whole.
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

template <int K, int NB, int UN, int DA, int OCC>
__global__ void __launch_bounds__(NT, OCC) k_conv_stream(ConvP p) {
  constexpr int COUT = NB * 32, CIN = UN * 32;
  constexpr int BROW = CIN * 2 + 16;                  // LDS pitch of a weight row (one output channel, one tap): +16 B pad =>
                                                      // ds_read_b128 of 16 different rows at one column is conflict-free
  constexpr int BSLOTS = UN * 4;                      // 16-B vectors per weight row
  constexpr int BVEC = COUT * BSLOTS;                 // 16-B vectors per tap
  constexpr int BPT = (BVEC + NT - 1) / NT;           // vectors per thread per tap
  constexpr int EP = 32 + 4;                          // epilogue pitch (floats), one 32-column block at a time
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Bs = smem;                                                          // [2][COUT][BROW]
  float* Es = reinterpret_cast<float*>(smem + 2 * (size_t)COUT * BROW);     // [WAVES][32][EP]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  const int tile = xcd_tile(blockIdx.x, p.nblk);
  const int64_t r0 = (int64_t)tile * (WAVES * 32) + wv * 32;
  const int64_t row = r0 + fi;
  const bool rvalid = row < p.n_out;

  int idx[K];
#pragma unroll
  for (int k = 0; k < K; ++k) idx[k] = rvalid ? (p.table ? p.table[(int64_t)k * p.n_out + row] : (int)row) : -1;

  const int in_ld_b = (int)(p.in_ld * 2);
  const int64_t in_bytes = ((int64_t)p.n_in - 1) * in_ld_b + (int64_t)CIN * 2;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)in_bytes, 0x00020000);
  const unsigned lane_off = (unsigned)(fh * 16);
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

  u32x4 a[DA][UN][2];
  u32x4 bw[BPT];
  auto issue_a = [&](int k, u32x4 (&dst)[UN][2]) __attribute__((always_inline)) {
    const unsigned base = (unsigned)idx[k] * (unsigned)in_ld_b + lane_off;
#pragma unroll
    for (int c = 0; c < UN; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        dst[c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + c * 64 + j * 32), 0, 0));
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
      for (int j = 0; j < 2; ++j) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, a[k % DA][c][j]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int slot = c * 4 + 2 * j + fh;
          const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bl + nb * 32 * BROW + slot * 16);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[nb], 0, 0, 0);
        }
      }
    if (k + DA < K) issue_a(k + DA, a[k % DA]);
    if (k + 1 < K) __syncthreads();
  }

  // epilogue, one 32-column block at a time through a wave-private LDS transposition buffer
  float* ew = Es + wv * 32 * EP;
  const char* res = (const char*)p.res;
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
        const int c0 = nb * 32 + cvv * 8;
        if (res) {
          const u32x4 rv = *reinterpret_cast<const u32x4*>(res + (orow * p.res_ld + c0) * 2);
#pragma unroll
          for (int q = 0; q < 4; ++q) { v[2 * q] += bf16_lo(rv[q]); v[2 * q + 1] += bf16_hi(rv[q]); }
        }
        epi_store8<true>(p.out, p.out_ld, p.out_scale, p.out_shift, p.out_relu, orow, c0, v);
        if (p.out2) epi_store8<true>(p.out2, p.out2_ld, p.out2_scale, p.out2_shift, p.out2_relu, orow, c0, v);
        if (p.out3) epi_store8<true>(p.out3, p.out3_ld, p.out3_scale, p.out3_shift, p.out3_relu, orow, c0, v);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int K, int NB, int UN, int DA>
int launch(ConvP p, hipStream_t s) {
  constexpr int OCC = (NB * 16 + DA * UN * 8 + 40 <= 118) ? 4 : 2;      // rough VGPR need -> waves per SIMD to ask for
  const size_t lds = 2 * (size_t)NB * 32 * (UN * 64 + 16) + (size_t)WAVES * 32 * 36 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_stream<K, NB, UN, DA, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return TL_ERR_LAUNCH;
    attr_set = true;
  }
  p.nblk = (int)tl_cdiv(p.n_out, WAVES * 32);
  k_conv_stream<K, NB, UN, DA, OCC><<<p.nblk, NT, lds, s>>>(p);
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

template <int K>
int dispatch(const ConvP& p, hipStream_t s) {
  const int nb = p.Cout / 32, un = p.Cin / 32;
#define TL_S(NB_, UN_, DA_) if (nb == NB_ && un == UN_) return launch<K, NB_, UN_, DA_>(p, s);
  TL_S(2, 2, 3) TL_S(2, 4, 2) TL_S(3, 3, 2) TL_S(3, 6, 1) TL_S(4, 4, 2) TL_S(2, 3, 3) TL_S(3, 2, 3) TL_S(3, 4, 2) TL_S(4, 3, 2) TL_S(1, 2, 3) TL_S(2, 1, 3)
#undef TL_S
  return TL_ERR_UNSUPPORTED;
}

}  // namespace

int tl_launch_conv_stream(const ConvP& p, hipStream_t s) {
  if (p.in_scale || p.in_relu) return TL_ERR_UNSUPPORTED;
  const int64_t ld_b = p.in_ld * 2, in_bytes = (p.n_in - 1) * ld_b + (int64_t)p.Cin * 2;
  if (!(in_bytes > 0 && in_bytes + 2 * ld_b < 0xFFFFFFFFll)) return TL_ERR_UNSUPPORTED;
  switch (p.K) {
    case 27: return dispatch<27>(p, s);
    case 8: return dispatch<8>(p, s);
  }
  return TL_ERR_UNSUPPORTED;
}
