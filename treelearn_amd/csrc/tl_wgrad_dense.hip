// Weight gradient of the 27-tap convs of the big levels, "dense over taps" form (bf16 operands, fp32 accumulation):
//     gW[k][co][ci] = sum_o gout[o][co] * x[nbr[k][o]][ci]            (reference: spconv's autograd behind SubMConv3d.forward,
//                                                                      reached from tools/training/train.py:40 `.backward()`)
// The contraction index of this GEMM is the ROW, so both MFMA operands are needed "8 rows of one channel per lane": a transposed
// view of the row-major feature matrices.  tl_wgrad.hip compacts the present (output row, input row) pairs of a tap into a list and
// gathers BOTH operands per pair; the list look-ups, two gathers and two LDS transpositions per 16 pairs form one dependent chain
// per wave and the matrix pipes sit idle 90 % of the time.  Here nothing is compacted:
//   * a workgroup walks blocks of MBR consecutive output rows; their gout tile goes to LDS ONCE (coalesced, double-buffered, one
//     barrier per block) and is the A operand of every tap: its fragments are read with ds_read_b64_tr_b16, two reads per 32 channels;
//   * every WAVE owns one (tap, 64- or 32-channel slice of Cin) job and keeps that job's [Cout x slice] fp32 tile in registers for the
//     whole kernel; per step of 16 (32) rows it gathers the tap's input rows with full-row 16-B buffer loads straight through the
//     rulebook column -- an absent neighbour is index -1 = an out-of-range offset = zeros from the hardware, it simply contributes
//     nothing -- parks them in a wave-private LDS tile and reads the B fragments back transposed.  No pair lists, no per-pair gout
//     gather, no branches: the loop body is straight-line code with PD steps of gathers in flight (counted vmcnt);
//   * the matrix work is that of the dense form (27 / 16.4 of the present pairs at level 2, 27 / 5.5 at level 1) -- at most a quarter
//     of the kernel's time on this part, which is what buys the simple dataflow;
//   * LDS tiles are unpadded and XOR-swizzled by 16-B piece so that the 4 rows x 64 B a half-wave touches per transposing read fall
//     into the four different 64-B quarters of the 256-B bank row (swz<>), and the 16-B row writes stay whole 128-B lines;
//   * persistent workgroups (one per CU): workgroup bx owns the macro blocks {xcd * M/8 + slot + j * slots} -- every XCD a contiguous
//     eighth of the rows (neighbouring rows gather overlapping input rows: one L2), interleaved inside the XCD -- and writes ONE
//     partial tile set; k_wgrad_reduce_par adds the <= 128 partials in a fixed order.  Deterministic.
#include "tl_conv_internal.h"
#include "tl_f16_train.h"
#include <atomic>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4* lds4;

// 16-B piece p of row r of a tile with row pitch P bytes lives at piece slot p ^ swz<P>(r)
template <int P>
static __device__ __forceinline__ int swz(int r) {
  if constexpr ((P / 64) % 2 == 1) return 0;                       // odd multiple of 64 B: consecutive rows already rotate through the quarters
  else if constexpr ((P / 64) % 4 == 2) return ((r >> 1) & 1) << 2;
  else return (r & 3) << 2;
}

// NBO = Cout / 32; NBIW = 32-channel blocks of the wave's Cin slice (1 or 2); KS = 16-row MFMA steps per gather step;
// PD = gather steps in flight per wave (2 or 4); NW = waves per workgroup
template <int NBO, int NBIW, int KS, int PD, int NW>
__global__ void __launch_bounds__(NW * 64) k_wgrad_dense(const uint16_t* __restrict__ x, int64_t x_ld, const uint16_t* __restrict__ g, int64_t g_ld,
                                                         const int32_t* __restrict__ table, int64_t n_out, int64_t n_in, int K, int Cin, int NS,
                                                         int nmb, float* __restrict__ ws) {
  constexpr int NT = NW * 64;
  constexpr int COUT = NBO * 32, CS = NBIW * 32, PG = COUT * 2, PX = CS * 2, ROWS = 16 * KS, SPM = 4, MBR = ROWS * SPM;
  constexpr int PRG = COUT / 8, PRX = CS / 8;
  constexpr int NPT = (MBR * PRG + NT - 1) / NT;                   // 16-B pieces of the gout tile per thread
  constexpr int NL = ROWS * PRX / 64;                              // gather instructions per lane and step
  static_assert(ROWS * PRX % 64 == 0 && SPM % PD == 0, "configuration");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* Gs = smem;                                                 // [2][MBR][PG]
  char* Xs = smem + 2 * MBR * PG + wv * (2 * ROWS * PX);           // wave-private [2][ROWS][PX]

  const int job = (int)blockIdx.y * NW + wv;
  const int tap = job / NS, slice = job % NS;
  const bool active = tap < K;
  const int bx = (int)blockIdx.x, nslot = (int)gridDim.x >> 3, xcd = bx & 7, slot = bx >> 3;
  const int m8 = (nmb + 7) >> 3;
  const int m_lo = xcd * m8 + slot, m_hi = min(nmb, (xcd + 1) * m8);
  const int niter = m_lo < m_hi ? (m_hi - 1 - m_lo) / nslot + 1 : 0;
  auto mb_of = [&](int it) { return it < niter ? m_lo + it * nslot : -1; };

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, (int)min((int64_t)0x7FFFFFFF, n_in * x_ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g), 0, (int)min((int64_t)0x7FFFFFFF, n_out * g_ld * 2), 0x00020000);
  const unsigned x_ld_b = (unsigned)(x_ld * 2);

  // gather pattern: piece q = lane + 64 i of the step's [ROWS][PRX] pieces
  int xrow[NL]; unsigned xcol[NL]; int xlds[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int q = lane + 64 * i, r = q / PRX, pc = q % PRX;
    xrow[i] = r;
    xcol[i] = (unsigned)((slice * CS + pc * 8) * 2);
    xlds[i] = r * PX + ((pc ^ swz<PX>(r)) << 4);
  }
  // transposing reads: lane (fh, tg, ti) supplies the 8-byte chunk (row 8 fh + 4 q + (ti >> 2), channels 16 tg + 4 (ti & 3) ..) of a 32-channel block
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

  // rulebook entries of a macro block: lane l holds rows l, l + 64, ... of the block
  auto idx_load = [&](int it, int (&dst)[KS]) __attribute__((always_inline)) {
    const int mb = mb_of(it);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int64_t row = (int64_t)mb * MBR + kk * 64 + lane;
      dst[kk] = (mb >= 0 && active && row < n_out) ? table[(int64_t)tap * n_out + row] : -1;
    }
  };
  auto g_load = [&](int it, u32x4 (&dst)[NPT]) __attribute__((always_inline)) {
    const int mb = mb_of(it);
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int q = tid + i * NT, r = q / PRG, pc = q % PRG;
      const int64_t row = (int64_t)mb * MBR + r;
      const bool ok = mb >= 0 && row < n_out && (NPT * NT == MBR * PRG || q < MBR * PRG);
      const unsigned off = ok ? (unsigned)(row * g_ld * 2) + (unsigned)(pc * 16) : 0xFFFFFFFFu;
      dst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)off, 0, 0));
    }
  };
  auto g_store = [&](int buf, const u32x4 (&src)[NPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int q = tid + i * NT, r = q / PRG, pc = q % PRG;
      if (NPT * NT == MBR * PRG || q < MBR * PRG) *reinterpret_cast<u32x4*>(Gs + buf * (MBR * PG) + r * PG + ((pc ^ swz<PG>(r)) << 4)) = src[i];
    }
  };
  // gathers of step s (0 .. SPM-1) of a macro block whose rulebook entries are `idx`
  auto x_issue = [&](int s, const int (&idx)[KS], u32x4 (&dst)[NL]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int rmb = s * ROWS + xrow[i];                          // row inside the macro block; its entry sits in lane rmb & 63 of idx[rmb >> 6]
      int e;
      if constexpr (KS == 1) e = __builtin_amdgcn_ds_bpermute((rmb & 63) << 2, idx[0]);
      else {
        // ROWS = 32, MBR = 128: steps 0, 1 read idx[0], steps 2, 3 idx[1] (s is a compile-time constant after unrolling)
        e = __builtin_amdgcn_ds_bpermute((rmb & 63) << 2, idx[(s * ROWS) >> 6]);
      }
      const unsigned off = (unsigned)e * x_ld_b + xcol[i];          // e = -1 wraps past the end of the buffer: zeros
      dst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0));
    }
  };

  if (niter > 0) {                                                  // uniform over the workgroup (an idle workgroup still writes its zero partials)
  int idx0[KS], idx1[KS], idx2[KS];
  idx_load(0, idx0); idx_load(1, idx1); idx_load(2, idx2);
  {
    u32x4 g0[NPT];
    g_load(0, g0);
    g_store(0, g0);
  }
  u32x4 xb[PD][NL];
#pragma unroll
  for (int d = 0; d < PD; ++d) x_issue(d, idx0, xb[d]);
  __syncthreads();

  for (int it = 0; it < niter; ++it) {
    u32x4 gn[NPT];
    g_load(it + 1, gn);
    const char* Gc = Gs + (it & 1) * (MBR * PG);
#pragma unroll
    for (int s = 0; s < SPM; ++s) {
      char* Xc = Xs + (s & 1) * (ROWS * PX);
#pragma unroll
      for (int i = 0; i < NL; ++i) *reinterpret_cast<u32x4*>(Xc + xlds[i]) = xb[s % PD][i];
      // the slot is free again: gathers of the step PD ahead (the next macro block's entries from step SPM - PD on)
      if (s + PD < SPM) x_issue(s + PD, idx0, xb[s % PD]);
      else x_issue(s + PD - SPM, idx1, xb[s % PD]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u32x4 A[NBO], B[NBIW];
#pragma unroll
        for (int a = 0; a < NBO; ++a)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(Gc + (s * ROWS + 16 * ks + 4 * q) * PG + ga[a])));
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
    g_store((it + 1) & 1, gn);                                      // every wave passed the previous barrier after its last read of that buffer
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) { idx0[kk] = idx1[kk]; idx1[kk] = idx2[kk]; }
    idx_load(it + 3, idx2);
    __syncthreads();
  }
  }

  if (!active) return;
  float* wp = ws + (((int64_t)bx * K + tap) * COUT) * (int64_t)Cin + slice * CS;
#pragma unroll
  for (int a = 0; a < NBO; ++a)
#pragma unroll
    for (int b = 0; b < NBIW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh, ci = b * 32 + fi;
        wp[(int64_t)co * Cin + ci] = acc[a][b][r];
      }
}

// The same kernel with the gathers staged by LDS-DMA (`buffer_load ... lds`: global -> LDS without passing through registers).  PMC on the
// register-staged form (profiles/r3_wgrad/): no LDS bank conflicts, MFMA pipes busy 31 %, waves parked 44 % of their cycles, LDS
// instructions more than half of what a wave issues -- and 9 waves of 142 VGPRs per CU with 8 KB of gathers in flight each, i.e. the
// kernel is bound by how many bytes are in flight against the L2 round trip.  Here a wave owns a RING of NR row tiles in LDS: the DMA of
// step t + NR - 1 is issued at step t into the slot step t - 1 just read (no staging registers: 14 waves of <= 128 VGPRs per CU, and no
// ds_write instructions), the source-side swizzle (lane L fetches piece (L % PRX) ^ swz(row) so that the contiguous 1 KB a DMA
// instruction writes IS the swizzled tile), counted `s_waitcnt vmcnt(N)` with N fixed per unrolled step (vmcnt retires in order; the gout /
// rulebook loads of a block start are part of the count), raw `s_barrier` (a `__syncthreads()` would drain the DMAs).  Absent neighbours:
// the out-of-range lanes of a DMA write zeros.  Same results bit for bit (same summation order).
template <int NBO, int NBIW, int KS, int NR, int NW>
__global__ void __launch_bounds__(NW * 64) k_wgrad_dense_dma(const uint16_t* __restrict__ x, int64_t x_ld, const uint16_t* __restrict__ g, int64_t g_ld,
                                                             const int32_t* __restrict__ table, int64_t n_out, int64_t n_in, int K, int Cin, int NS,
                                                             int nmb, float* __restrict__ ws) {
  constexpr int NT = NW * 64;
  constexpr int COUT = NBO * 32, CS = NBIW * 32, PG = COUT * 2, PX = CS * 2, ROWS = 16 * KS, SPM = 4, MBR = ROWS * SPM;
  constexpr int PRG = COUT / 8, PRX = CS / 8;
  constexpr int NPT = (MBR * PRG + NT - 1) / NT;
  constexpr int NL = ROWS * PRX / 64;
  constexpr int SLOT = ROWS * PX;
  static_assert(ROWS * PRX % 64 == 0 && SLOT == NL * 1024 && NR >= 2 && NR <= 9, "configuration");
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* Gs = smem;                                                 // [2][MBR][PG]
  char* Xs = smem + 2 * MBR * PG + wv * (NR * SLOT);               // wave-private ring [NR][ROWS][PX]

  const int job = (int)blockIdx.y * NW + wv;
  const int tap = job / NS, slice = job % NS;
  const bool active = tap < K;
  const int bx = (int)blockIdx.x, nslot = (int)gridDim.x >> 3, xcd = bx & 7, slot = bx >> 3;
  const int m8 = (nmb + 7) >> 3;
  const int m_lo = xcd * m8 + slot, m_hi = min(nmb, (xcd + 1) * m8);
  const int niter = m_lo < m_hi ? (m_hi - 1 - m_lo) / nslot + 1 : 0;
  auto mb_of = [&](int it) { return it < niter ? m_lo + it * nslot : -1; };

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, (int)min((int64_t)0x7FFFFFFF, n_in * x_ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g), 0, (int)min((int64_t)0x7FFFFFFF, n_out * g_ld * 2), 0x00020000);
  const unsigned x_ld_b = (unsigned)(x_ld * 2);

  // DMA pattern: instruction i writes LDS bytes [1024 i + 16 lane, +16) of the slot = tile row q / PRX, piece slot q % PRX (q = lane + 64 i),
  // which must hold piece (q % PRX) ^ swz(row) of the gathered row
  int xrow[NL]; unsigned xcol[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int q = lane + 64 * i, r = q / PRX, ps = q % PRX;
    xrow[i] = r;
    xcol[i] = (unsigned)((slice * CS + (ps ^ swz<PX>(r)) * 8) * 2);
  }
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

  // (bounds-checked buffer loads, NOT predicated global loads: a load the compiler may branch around when no lane needs it would make the
  // number of outstanding memory operations -- which the counted waits below rely on -- depend on the data: at the end of a wave's range
  // the waits then came too early and the step read a slot its DMA had not filled yet)
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(table + (active ? (int64_t)tap * n_out : 0)), 0,
                                                                     (int)min((int64_t)0x7FFFFFFF, n_out * 4), 0x00020000);
  auto idx_load = [&](int it, int (&dst)[KS]) __attribute__((always_inline)) {
    const int mb = mb_of(it);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int64_t row = (int64_t)mb * MBR + kk * 64 + lane;
      const bool ok = mb >= 0 && active && row < n_out;
      const int v = __builtin_amdgcn_raw_buffer_load_b32(rt, ok ? (int)(row * 4) : -1, 0, 0);
      dst[kk] = ok ? v : -1;
    }
  };
  // the same two loads for the main loop, as inline assembly: their POSITION in the instruction stream is part of the wait arithmetic (hipcc
  // sank the compiler-visible gout load from the block start to its use behind four steps of DMAs and drained them there); the loaded
  // registers are handed to the compiler only after the counted wait that covers them.  A row past the end is offset -1 = out of range =
  // 0 from the hardware, which is a valid row index: validity is re-applied after the wait (idx_fix).
  auto idx_load_asm = [&](int it, int (&dst)[KS]) __attribute__((always_inline)) {
    const int mb = mb_of(it);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int64_t row = (int64_t)mb * MBR + kk * 64 + lane;
      const bool ok = mb >= 0 && active && row < n_out;
      const int off = ok ? (int)(row * 4) : -1;
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(dst[kk]) : "v"(off), "s"(rt));
    }
  };
  auto idx_fix = [&](int it, int (&dst)[KS]) __attribute__((always_inline)) {      // after the wait: rows past the end read 0 -> absent
    const int mb = mb_of(it);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      asm volatile("" : "+v"(dst[kk]));
      const int64_t row = (int64_t)mb * MBR + kk * 64 + lane;
      if (!(mb >= 0 && active && row < n_out)) dst[kk] = -1;
    }
  };
  auto g_load_asm = [&](int it, u32x4 (&dst)[NPT]) __attribute__((always_inline)) {
    const int mb = mb_of(it);
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int q = tid + i * NT, r = q / PRG, pc = q % PRG;
      const int64_t row = (int64_t)mb * MBR + r;
      const bool ok = mb >= 0 && row < n_out && (NPT * NT == MBR * PRG || q < MBR * PRG);
      const unsigned off = ok ? (unsigned)(row * g_ld * 2) + (unsigned)(pc * 16) : 0xFFFFFFFFu;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst[i]) : "v"(off), "s"(rg));
    }
  };
  auto g_load = [&](int it, u32x4 (&dst)[NPT]) __attribute__((always_inline)) {
    const int mb = mb_of(it);
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int q = tid + i * NT, r = q / PRG, pc = q % PRG;
      const int64_t row = (int64_t)mb * MBR + r;
      const bool ok = mb >= 0 && row < n_out && (NPT * NT == MBR * PRG || q < MBR * PRG);
      const unsigned off = ok ? (unsigned)(row * g_ld * 2) + (unsigned)(pc * 16) : 0xFFFFFFFFu;
      dst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)off, 0, 0));
    }
  };
  auto g_store = [&](int buf, const u32x4 (&src)[NPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int q = tid + i * NT, r = q / PRG, pc = q % PRG;
      if (NPT * NT == MBR * PRG || q < MBR * PRG) *reinterpret_cast<u32x4*>(Gs + buf * (MBR * PG) + r * PG + ((pc ^ swz<PG>(r)) << 4)) = src[i];
    }
  };
  // DMA of step s (0 .. SPM-1) of a macro block whose rulebook entries are `idx` into ring slot `rs`
  auto x_dma = [&](int s, const int (&idx)[KS], int rs) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int rmb = s * ROWS + xrow[i];
      const int e = __builtin_amdgcn_ds_bpermute((rmb & 63) << 2, idx[(s * ROWS) >> 6]);
      const unsigned off = (unsigned)e * x_ld_b + xcol[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(Xs + rs * SLOT + i * 1024), 16, (int)off, 0, 0, 0);
    }
  };
  // VMEM operations younger than the DMA of the step being consumed, per position s in the macro block: the DMAs of the NR - 1 steps
  // issued since, plus the (gout, rulebook) loads of every block start among them (issued BEFORE that step's DMA)
  auto younger = [](int s) constexpr {
    int c = 0;
    for (int j = 0; j <= NR - 2; ++j) if (((s - j) % SPM + SPM) % SPM == 0) ++c;
    return (NR - 1) * NL + c * (NPT + KS);
  };

  if (niter > 0) {
  int idx0[KS], idx1[KS], idx2[KS], idx3[KS];
  idx_load(0, idx0); idx_load(1, idx1); idx_load(2, idx2);
  {
    u32x4 g0[NPT];
    g_load(0, g0);
    g_store(0, g0);
  }
  int head = 0;                                                    // ring slot of the step about to be consumed
  const unsigned gs_off = (unsigned)(uintptr_t)(lds_ptr)Gs, xs_off = (unsigned)(uintptr_t)(lds_ptr)Xs;
#pragma unroll
  for (int d = 0; d < NR - 1; ++d) {                               // steps 0 .. NR-2 of the first blocks
    if (d / SPM == 0) x_dma(d % SPM, idx0, d);
    else if (d / SPM == 1) x_dma(d % SPM, idx1, d);
    else x_dma(d % SPM, idx2, d);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (one-time drain: from here on every wait is counted)

  for (int it = 0; it < niter; ++it) {
    u32x4 gn[NPT];
    const unsigned gc_off = gs_off + (unsigned)((it & 1) * (MBR * PG));
#pragma unroll
    for (int s = 0; s < SPM; ++s) {
      if (s == 0) { g_load_asm(it + 1, gn); idx_load_asm(it + 3, idx3); }
      // the slot read in the previous step is free (its fragments fed MFMAs that were issued already): DMA of step s + NR - 1
      {
        constexpr int dummy = 0; (void)dummy;
        const int sn = s + NR - 1;
        const int rs = head == 0 ? NR - 1 : head - 1;
        if (sn / SPM == 0) x_dma(sn % SPM, idx0, rs);
        else if (sn / SPM == 1) x_dma(sn % SPM, idx1, rs);
        else x_dma(sn % SPM, idx2, rs);
      }
      __builtin_amdgcn_sched_barrier(0);
      switch (s) {                                                 // everything up to this step's DMA has landed
        case 0: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger(0)) : "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger(1)) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger(2)) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger(3)) : "memory"); break;
      }
      __builtin_amdgcn_sched_barrier(0);
      // the fragment reads are inline assembly: a DS read the compiler can see is preceded by `s_waitcnt vmcnt(0)` while any LDS-DMA
      // is outstanding (it cannot tell the slots apart), which would serialise the ring
      const unsigned xb = xs_off + (unsigned)(head * SLOT);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u32x2 Ah[NBO][2], Bh[NBIW][2];
#pragma unroll
        for (int a = 0; a < NBO; ++a) {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(Ah[a][0]) : "v"(gc_off + (unsigned)ga[a]), "n"((s * ROWS + 16 * ks) * PG));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(Ah[a][1]) : "v"(gc_off + (unsigned)ga[a]), "n"((s * ROWS + 16 * ks + 4) * PG));
        }
#pragma unroll
        for (int b = 0; b < NBIW; ++b) {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(Bh[b][0]) : "v"(xb + (unsigned)xa[b]), "n"((16 * ks) * PX));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(Bh[b][1]) : "v"(xb + (unsigned)xa[b]), "n"((16 * ks + 4) * PX));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        u32x4 A[NBO], B[NBIW];
#pragma unroll
        for (int a = 0; a < NBO; ++a) {
          asm volatile("" : "+v"(Ah[a][0]), "+v"(Ah[a][1]));       // orders the consumers after the wait
          A[a][0] = Ah[a][0][0]; A[a][1] = Ah[a][0][1]; A[a][2] = Ah[a][1][0]; A[a][3] = Ah[a][1][1];
        }
#pragma unroll
        for (int b = 0; b < NBIW; ++b) {
          asm volatile("" : "+v"(Bh[b][0]), "+v"(Bh[b][1]));
          B[b][0] = Bh[b][0][0]; B[b][1] = Bh[b][0][1]; B[b][2] = Bh[b][1][0]; B[b][3] = Bh[b][1][1];
        }
#pragma unroll
        for (int a = 0; a < NBO; ++a)
#pragma unroll
          for (int b = 0; b < NBIW; ++b)
            acc[a][b] = h16_mfma(A[a], B[b], acc[a][b]);
      }
      __builtin_amdgcn_sched_barrier(0);
      head = head + 1 == NR ? 0 : head + 1;
    }
    // the block-start loads are older than the SPM * NL gathers issued since
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPM * NL) : "memory");
#pragma unroll
    for (int i = 0; i < NPT; ++i) asm volatile("" : "+v"(gn[i]));
    idx_fix(it + 3, idx3);
    g_store((it + 1) & 1, gn);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) { idx0[kk] = idx1[kk]; idx1[kk] = idx2[kk]; idx2[kk] = idx3[kk]; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the DMAs issued past the end (all out of range) before the slots' memory is given back
  }

  if (!active) return;
  float* wp = ws + (((int64_t)bx * K + tap) * COUT) * (int64_t)Cin + slice * CS;
#pragma unroll
  for (int a = 0; a < NBO; ++a)
#pragma unroll
    for (int b = 0; b < NBIW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh, ci = b * 32 + fi;
        wp[(int64_t)co * Cin + ci] = acc[a][b][r];
      }
}

// gw[e] = sum over the partial tile sets.  A workgroup = 16 float4 elements x 16 part lanes: lane pl adds parts pl, pl + 16, ... (up to
// 8 loads, all in flight together), the sixteen lane sums are added in lane order through LDS: deterministic, and the partials -- just
// written by the weight-gradient kernel -- stream out of L2 / Infinity Cache with thousands of loads in flight instead of one chain
// per element.
// KR > 0: gw is written in the REFERENCE parameter layout [Cout][K][Cin] (ws / the default output are [K][Cout][Cin]); Cin % 4 == 0
__global__ void __launch_bounds__(256) k_wgrad_reduce_par(const float* __restrict__ ws, int nparts, int64_t per, float* __restrict__ gw, int KR, int CoutR, int CinR) {
  __shared__ f32x4 red[16][16];
  const int e = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int64_t v = (int64_t)blockIdx.x * 16 + e, stride = per / 4;
  f32x4 s = {0, 0, 0, 0};
  if (v < stride) {
    const f32x4* src = reinterpret_cast<const f32x4*>(ws) + v;
    for (int p0 = pl; p0 < nparts; p0 += 128) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int p = p0 + 16 * u; t[u] = p < nparts ? src[(int64_t)p * stride] : f32x4{0, 0, 0, 0}; }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
  }
  red[pl][e] = s;
  __syncthreads();
  if (pl == 0 && v < stride) {
    f32x4 a = red[0][e];
#pragma unroll
    for (int l = 1; l < 16; ++l) a += red[l][e];
    int64_t o = v * 4;
    if (KR > 0) {
      const int ci = (int)(o % CinR); const int64_t r = o / CinR; const int co = (int)(r % CoutR); const int k = (int)(r / CoutR);
      o = ((int64_t)co * KR + k) * CinR + ci;
    }
    *reinterpret_cast<f32x4*>(gw + o) = a;
  }
}

template <int NBO, int NBIW, int KS, int PD, int NW>
int launch(const uint16_t* x, int64_t x_ld, const uint16_t* g, int64_t g_ld, const int32_t* table, int64_t n_out, int64_t n_in, int K, int Cin, int NS,
           int gx, float* ws, hipStream_t s) {
  constexpr int MBR = 64 * KS;
  const size_t lds = 2 * (size_t)MBR * NBO * 64 + (size_t)NW * 2 * (16 * KS) * (NBIW * 64);
  static TlAttrOnce attr_once;                     // per kernel instantiation AND device (the attribute is per device)
  if (!tl_lds_attr(attr_once, reinterpret_cast<const void*>(&k_wgrad_dense<NBO, NBIW, KS, PD, NW>), 160 * 1024)) return TL_ERR_LAUNCH;
  const int nmb = (int)tl_cdiv(n_out, MBR);
  const int gy = (int)tl_cdiv((int64_t)K * NS, NW);
  k_wgrad_dense<NBO, NBIW, KS, PD, NW><<<dim3((unsigned)gx, (unsigned)gy), NW * 64, lds, s>>>(x, x_ld, g, g_ld, table, n_out, n_in, K, Cin, NS, nmb, ws);
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

template <int NBO, int NBIW, int KS, int NR, int NW>
int launch_dma(const uint16_t* x, int64_t x_ld, const uint16_t* g, int64_t g_ld, const int32_t* table, int64_t n_out, int64_t n_in, int K, int Cin, int NS,
               int gx, float* ws, hipStream_t s) {
  constexpr int MBR = 64 * KS;
  const size_t lds = 2 * (size_t)MBR * NBO * 64 + (size_t)NW * NR * (16 * KS) * (NBIW * 64);
  if (lds > 160 * 1024) return TL_ERR_UNSUPPORTED;
  static TlAttrOnce attr_once;                     // per kernel instantiation AND device (the attribute is per device)
  if (!tl_lds_attr(attr_once, reinterpret_cast<const void*>(&k_wgrad_dense_dma<NBO, NBIW, KS, NR, NW>), 160 * 1024)) return TL_ERR_LAUNCH;
  const int nmb = (int)tl_cdiv(n_out, MBR);
  const int gy = (int)tl_cdiv((int64_t)K * NS, NW);
  k_wgrad_dense_dma<NBO, NBIW, KS, NR, NW><<<dim3((unsigned)gx, (unsigned)gy), NW * 64, lds, s>>>(x, x_ld, g, g_ld, table, n_out, n_in, K, Cin, NS, nmb, ws);
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// Level 1, block-local rows (tl_blk): the weight gradient of a 27-tap 32 -> 32 SubM conv in the staged-unit form of the forward kernel
// (tl_conv_blk.hip).  The dense-over-taps kernel above gathers the input rows of every tap from memory -- 27 gathers per output row where 5.5
// neighbours exist; here a workgroup walks UNITS (<= 64 own rows + their <= 126 halo rows, 1.8 staged rows per own row): the unit's x rows, its
// gout rows and its local rulebook go to LDS once, and all 27 taps contract against them:
//     gW[k][co][ci] += sum over the unit's rows r of gout[r][co] * X[pos(r, k)][ci]          pos = the 10-bit local rulebook entry >> 2, 191 = absent
// Wave w owns taps 3w, 3w + 1, 3w + 2 -- exactly the three entries of word w of a row's nine rulebook words -- and keeps their three 32 x 32 fp32
// tiles in registers for the whole kernel.  Per 16-row step: two transposing reads give the gout fragments (shared by the three taps), two reads the
// rulebook words of the lane's two rows, and per tap two transposing reads AT THE NEIGHBOURS' staged positions (every lane supplies its row's
// address) give the x fragments -- absent neighbours point at a zero row.  Staging goes through two register sets and two LDS buffers: while unit u
// is contracted, the rows of unit u + 1 are landing and those of unit u + 2 (and the halo indices of unit u + 3) are requested; one barrier per unit.  Persistent workgroups, unit i -> workgroup
// i mod G; every workgroup writes ONE partial set, k_wgrad_reduce_par adds them in workgroup order: deterministic.
constexpr int kWB_NW = 9, kWB_NT = kWB_NW * 64;
constexpr int kWB_XB = 192 * 64, kWB_GB = 64 * 64, kWB_LB = 64 * 9 * 4, kWB_BUF = kWB_XB + kWB_GB + kWB_LB;       // bytes per buffer: 12288 + 4096 + 2304
constexpr int kWB_PARTS = 512;

__global__ void __launch_bounds__(kWB_NT) k_wgrad_blk(const uint16_t* __restrict__ x, int64_t x_ld, const uint16_t* __restrict__ g, int64_t g_ld,
                                                      const int32_t* __restrict__ unit, const int32_t* __restrict__ counter, const int32_t* __restrict__ halo,
                                                      const uint32_t* __restrict__ lrb, int64_t n, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][X 192 rows x 64 B | G 64 rows x 64 B | L 64 rows x 9 words]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nu = counter[0];
  const int G = (int)gridDim.x, wg = (int)blockIdx.x;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, (int)min((int64_t)0x7FFFFFFF, n * x_ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g), 0, (int)min((int64_t)0x7FFFFFFF, n * g_ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(lrb), 0, (int)min((int64_t)0x7FFFFFFF, n * 36), 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(halo), 0, (int)min((int64_t)0x7FFFFFFF, n * 128), 0x00020000);
  // staged rows 190 and 191 of both buffers stay zero (191 = the row of absent neighbours); the pieces below cover positions 0..189
  for (int e = tid; e < 2 * 32; e += kWB_NT) {
    const int b = e >> 5, w = e & 31;
    reinterpret_cast<uint32_t*>(smem + b * kWB_BUF + 190 * 64)[w] = 0;
  }
  // this thread's pieces of a unit: X pieces p = tid and tid + 576 (position p / 4, 16-B piece p % 4; p < 760), one G piece (tid < 256), one word
  const int xp0 = tid, xp1 = tid + kWB_NT;
  const int pos0 = xp0 >> 2, pos1 = xp1 >> 2;                       // pos0 in 0..143, pos1 in 144..189 (tid < 184)
  const bool has1 = xp1 < 760;
  auto unit_of = [&](int it) { const int u = wg + it * G; return u < nu ? u : -1; };

  auto desc = [&](int it) { const int u = unit_of(it); return u >= 0 ? reinterpret_cast<const int4*>(unit)[u] : make_int4(0, 0, 0, 0); };
  auto halo_load = [&](const int4& d, int& h0, int& h1) __attribute__((always_inline)) {
    const int j0 = pos0 - 64, j1 = pos1 - 64;
    h0 = __builtin_amdgcn_raw_buffer_load_b32(rh, (pos0 >= 64 && j0 < d.z) ? (int)(((int64_t)d.x * 32 + j0) * 4) : -1, 0, 0);
    h1 = __builtin_amdgcn_raw_buffer_load_b32(rh, (has1 && j1 < d.z) ? (int)(((int64_t)d.x * 32 + j1) * 4) : -1, 0, 0);
    if (!(pos0 >= 64 && j0 < d.z)) h0 = -1;
    if (!(has1 && j1 < d.z)) h1 = -1;
  };
  // two register sets: while unit u is contracted, the rows of unit u + 1 are landing in one set and those of unit u + 2 are requested into the other
  // (a unit's loads then have two contractions and a barrier to arrive: one was not enough, the loop ran at one memory latency per unit)
  u32x4 xq0[2], xq1[2], gq[2]; uint32_t lw[2];
  auto rows_load = [&](const int4& d, int h0, int h1, int S) __attribute__((always_inline)) {
    // own rows: position < count -> row lo + position; halo rows through the list; everything else reads zeros (offset -1 = out of range)
    const int r0 = pos0 < 64 ? (pos0 < d.y ? d.x + pos0 : -1) : h0;
    const int r1 = has1 ? h1 : -1;
    xq0[S] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, r0 >= 0 ? (int)((unsigned)((int64_t)r0 * x_ld * 2) + (unsigned)((xp0 & 3) * 16)) : -1, 0, 0));
    xq1[S] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, r1 >= 0 ? (int)((unsigned)((int64_t)r1 * x_ld * 2) + (unsigned)((xp1 & 3) * 16)) : -1, 0, 0));
    const int gr = tid >> 2;
    gq[S] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (tid < 256 && gr < d.y) ? (int)((unsigned)((int64_t)(d.x + gr) * g_ld * 2) + (unsigned)((tid & 3) * 16)) : -1, 0, 0));
    const int lr = tid / 9;                                        // 576 words = 64 rows x 9
    lw[S] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rl, lr < d.y ? (int)(((int64_t)d.x * 9 + tid) * 4) : -1, 0, 0);
    if (!(lr < d.y)) lw[S] = 0x2FFBFEFFu;                          // three entries "absent" (191 * 4 + 3 = 767 each): rows past the unit's end contribute nothing
  };
  auto rows_store = [&](char* buf, int S) __attribute__((always_inline)) {
    *reinterpret_cast<u32x4*>(buf + xp0 * 16) = xq0[S];
    if (has1) *reinterpret_cast<u32x4*>(buf + xp1 * 16) = xq1[S];
    if (tid < 256) *reinterpret_cast<u32x4*>(buf + kWB_XB + tid * 16) = gq[S];
    *reinterpret_cast<uint32_t*>(buf + kWB_XB + kWB_GB + tid * 4) = lw[S];
  };

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const int fh = lane >> 5, ti = lane & 15, tg = (lane >> 4) & 1;
  const int prow = 8 * fh + (ti >> 2), pcolb = 32 * tg + 8 * (ti & 3);      // row of the 16-row step / byte column this lane addresses in a transposing read

  // All four 16-row steps of a unit, unrolled: the rulebook words and the gout fragments of every step are requested up front, the x fragments
  // follow as their positions arrive -- the reads are independent, so the matrix instructions of one step run while the next steps' operands are on
  // their way (with a loop over the steps every step paid two dependent LDS round trips: waves parked 51 % of their cycles).  Rows past the unit's
  // end are zeros in the gout tile and "absent" in the rulebook words: a short unit costs four steps like a full one (1.5 % of the units).
  auto contract = [&](const char* cur) __attribute__((always_inline)) {
    const char* Xc = cur; const char* Gc = cur + kWB_XB; const uint32_t* Lc = reinterpret_cast<const uint32_t*>(cur + kWB_XB + kWB_GB);
    uint32_t w2[4][2];
    u32x4 A[4];
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int q = 0; q < 2; ++q) w2[st][q] = Lc[(16 * st + prow + 4 * q) * 9 + wv];
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(Gc + (16 * st + prow + 4 * q) * 64 + pcolb)));
        A[st][2 * q] = v[0]; A[st][2 * q + 1] = v[1];
      }
    // (skipping the (step, tap) pairs without a present neighbour among their 16 rows -- one ballot each -- was measured slower: 0.406 against 0.375 ms;
    // in the block order few such pairs exist, as the forward's tile statistics say)
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        u32x4 B;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int pos = (int)((w2[st][q] >> (10 * t)) & 1023u) >> 2;
          const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(Xc + pos * 64 + pcolb)));
          B[2 * q] = v[0]; B[2 * q + 1] = v[1];
        }
        acc[t] = h16_mfma(A[st], B, acc[t]);
      }
  };

  // prologue: unit 0 into buffer 0; the rows of unit 1 requested into set 1, the halo indices of unit 2 into slot 0
  int4 ud[4];                                                      // descriptors of units u .. u + 3 (rotated by hand below); u + 4 is requested a phase before its use
  int hh0[2], hh1[2];                                              // halo rows of this thread's two X pieces, per set
  ud[0] = desc(0);
  halo_load(ud[0], hh0[0], hh1[0]);
  rows_load(ud[0], hh0[0], hh1[0], 0);
  rows_store(smem, 0);
  ud[1] = desc(1);
  halo_load(ud[1], hh0[1], hh1[1]);
  rows_load(ud[1], hh0[1], hh1[1], 1);
  ud[2] = desc(2);
  halo_load(ud[2], hh0[0], hh1[0]);
  ud[3] = desc(3);
  __syncthreads();
  // phase for unit u (buffer u & 1 ready): set (u + 1) & 1 holds unit u + 1 (landing), set u & 1 is free and takes unit u + 2.  Every address a phase
  // needs was loaded a phase earlier (descriptor -> halo indices -> rows are three dependent loads: each gets a phase of its own)
  auto phase = [&](int u, int S) __attribute__((always_inline)) {            // S = u & 1, a constant after unrolling
    const int4 d4 = desc(u + 4);
    // (the index loads FIRST: memory operations retire in order, and the next phase starts by waiting for these indices -- behind the row loads
    // that wait would drain the rows too and leave them one phase instead of two)
    int n0, n1;
    halo_load(ud[3], n0, n1);
    rows_load(ud[2], hh0[S], hh1[S], S);                          // unit u + 2 (zeros when there is none)
    hh0[S ^ 1] = n0; hh1[S ^ 1] = n1;                             // (slot S ^ 1 held unit u + 1's indices: its rows were requested a phase ago)
    contract(smem + S * kWB_BUF);
    rows_store(smem + (S ^ 1) * kWB_BUF, S ^ 1);                  // unit u + 1 has landed by now
    ud[0] = ud[1]; ud[1] = ud[2]; ud[2] = ud[3]; ud[3] = d4;
    __syncthreads();
  };
  for (int u = 0; unit_of(u) >= 0; u += 2) {
    phase(u, 0);
    if (unit_of(u + 1) < 0) break;
    phase(u + 1, 1);
  }
  // partial tiles of this workgroup: ws[wg][k][co][ci]
  const int fi = lane & 31;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float* wp = ws + ((int64_t)wg * 27 + (3 * wv + t)) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) wp[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fi] = acc[t][r];
  }
}

}  // namespace

int g_wgrad_dma = 1;              // tl_set_tuning("wgrad_dma", 0): the register-staged form everywhere

// shared with tl_wgrad.hip: gw = ordered sum of `nparts` partial tile sets of `per` floats (per % 4 == 0, 16-B aligned)
int tl_launch_wgrad_reduce(const float* ws, int64_t nparts, int64_t per, float* gw, hipStream_t s, int K, int Cout, int Cin, int ref_layout) {
  const bool remap = ref_layout && K > 1 && Cin % 4 == 0;
  k_wgrad_reduce_par<<<(unsigned)tl_cdiv(per / 4, 16), 256, 0, s>>>(ws, (int)nparts, per, gw, remap ? K : 0, Cout, Cin);
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

int g_wgrad_dense = 1;            // tl_set_tuning("wgrad_dense", 0) restores the pair-list kernels of tl_wgrad.hip everywhere
int64_t g_wgrad_dense_min_rows = 60000;

// row slots (= partial tile sets) of the dense form for a shape, 0 if the shape is not served
int g_wgrad_dense_gx = 0;          // tl_set_tuning("wgrad_dense_gx", n): slot count override (experiments)

int tl_wgrad_dense_slots(int64_t n_out, int K, int Cin, int Cout) {
  if (!g_wgrad_dense || K != 27 || n_out < g_wgrad_dense_min_rows) return 0;
  const int key = Cin * 1000 + Cout;
  int gx = 0;
  if (g_wgrad_dma) {
    switch (key) {                                         // 14 waves of <= 128 registers per workgroup, one workgroup per CU
      case 32032: case 64032: gx = 128; break;             // 27 jobs: 2 workgroups per slot
      case 64064: gx = g_wgrad_dma == 2 ? 128 : 80; break; // 27 jobs: 3 workgroups of 9 waves per slot
      case 128064: gx = g_wgrad_dma == 2 ? 64 : 40; break; // 54 jobs: 6 workgroups of 9 waves per slot
      case 96096: gx = 40; break;                          // 81 jobs: 6 per slot
      case 192096: gx = 64; break;                         // 162 jobs: 12 per slot, 768 workgroups = 3 per CU
      case 128128: gx = 32; break;
      case 256128: gx = 16; break;
    }
  } else {
    switch (key) {
      case 32032: case 64032: gx = 128; break;             // 27 jobs: 2 workgroups of 14 waves per slot
      case 64064: gx = 80; break;                          // 27 jobs: 3 workgroups of 9 waves (170 registers each) per slot
      case 128064: gx = 40; break;                         // 54 jobs: 6 workgroups of 9 waves per slot
      case 96096: gx = 40; break;                          // 81 jobs (27 taps x 3 slices of 32 channels): 6 workgroups per slot
      case 192096: gx = 64; break;                         // 162 jobs: 12 workgroups per slot (768 = 3 per CU; 16 slots left a quarter of the CUs idle)
      case 128128: gx = 32; break;                         // 108 jobs (27 taps x 4 slices): 8 workgroups per slot
      case 256128: gx = 16; break;                         // 216 jobs: 16 workgroups per slot
    }
  }
  if (gx && g_wgrad_dense_gx) gx = g_wgrad_dense_gx & ~7;
  return gx;
}

int tl_launch_wgrad_dense(const uint16_t* x, int64_t x_ld, const uint16_t* g, int64_t g_ld, const int32_t* table, int64_t n_out, int64_t n_in, int K, int Cin,
                          int Cout, float* gw, float* ws, hipStream_t s, int ref_layout) {
  const int gx = tl_wgrad_dense_slots(n_out, K, Cin, Cout);
  if (!gx || !table) return TL_ERR_UNSUPPORTED;
  int rc = TL_ERR_UNSUPPORTED;
#define TL_DMA(NBO, NBIW, KS, NR, NW, NS) launch_dma<NBO, NBIW, KS, NR, NW>(x, x_ld, g, g_ld, table, n_out, n_in, K, Cin, NS, gx, ws, s)
  if (g_wgrad_dma) switch (Cin * 1000 + Cout) {            // 64 -> 32 measured no faster than the register-staged form: not here
    case 32032: rc = TL_DMA(1, 1, 2, 4, 14, 1); break;
    case 64064: rc = g_wgrad_dma == 2 ? TL_DMA(2, 2, 1, 4, 14, 1) : TL_DMA(2, 2, 1, 7, 9, 1); break;
    case 128064: rc = g_wgrad_dma == 2 ? TL_DMA(2, 2, 1, 4, 14, 2) : TL_DMA(2, 2, 1, 7, 9, 2); break;
    case 96096: rc = TL_DMA(3, 1, 2, 4, 14, 3); break;
    case 192096: rc = TL_DMA(3, 1, 2, 4, 14, 6); break;
    case 128128: rc = TL_DMA(4, 1, 1, 8, 14, 4); break;
    case 256128: rc = TL_DMA(4, 1, 1, 8, 14, 8); break;
  }
#undef TL_DMA
  if (rc == TL_ERR_UNSUPPORTED) switch (Cin * 1000 + Cout) {
    case 32032: rc = launch<1, 1, 2, 4, 14>(x, x_ld, g, g_ld, table, n_out, n_in, K, Cin, 1, gx, ws, s); break;
    case 64032: rc = launch<1, 2, 1, 4, 14>(x, x_ld, g, g_ld, table, n_out, n_in, K, Cin, 1, gx, ws, s); break;
    case 64064: rc = launch<2, 2, 1, 4, 9>(x, x_ld, g, g_ld, table, n_out, n_in, K, Cin, 1, gx, ws, s); break;
    case 128064: rc = launch<2, 2, 1, 4, 9>(x, x_ld, g, g_ld, table, n_out, n_in, K, Cin, 2, gx, ws, s); break;
    case 96096: rc = launch<3, 1, 2, 4, 14>(x, x_ld, g, g_ld, table, n_out, n_in, K, Cin, 3, gx, ws, s); break;
    case 192096: rc = launch<3, 1, 2, 4, 14>(x, x_ld, g, g_ld, table, n_out, n_in, K, Cin, 6, gx, ws, s); break;
    case 128128: rc = launch<4, 1, 1, 4, 14>(x, x_ld, g, g_ld, table, n_out, n_in, K, Cin, 4, gx, ws, s); break;
    case 256128: rc = launch<4, 1, 1, 4, 14>(x, x_ld, g, g_ld, table, n_out, n_in, K, Cin, 8, gx, ws, s); break;
  }
  if (rc != TL_OK) return rc;
  const int64_t per = (int64_t)K * Cout * Cin;
  return tl_launch_wgrad_reduce(ws, gx, per, gw, s, K, Cout, Cin, ref_layout);
}

extern "C" {

// Weight gradient of a 27-tap 32 -> 32 SubM conv over a block-local level: see k_wgrad_blk.  16-bit dtypes only (TL_ERR_UNSUPPORTED otherwise:
// the caller takes tl_conv_wgrad over the level's plain table); ws >= tl_conv_wgrad_blk_ws_floats() floats; gw [27][32][32] or, with
// ref_layout, [32][27][32] (the parameter's own layout).
int64_t tl_conv_wgrad_blk_ws_floats(void) { return (int64_t)kWB_PARTS * 27 * 1024; }

int tl_conv_wgrad_blk(const void* x, int64_t x_ld, const void* gout, int64_t g_ld, int dtype, const int32_t* blk_unit, const int32_t* blk_counter,
                      const int32_t* blk_halo, const uint32_t* blk_lrb, int64_t n, int Cin, int Cout, float* gw, int ref_layout, float* ws, tl_stream_t stream) {
  if (!x || !gout || !blk_unit || !blk_counter || !blk_halo || !blk_lrb || !gw || !ws || n <= 0 || x_ld < Cin || g_ld < Cout) return TL_ERR_ARG;
#ifndef TL_F16_BUILD
  if (dtype == TL_F16) return tl_conv_wgrad_blk_f16(x, x_ld, gout, g_ld, TL_BF16, blk_unit, blk_counter, blk_halo, blk_lrb, n, Cin, Cout, gw, ref_layout, ws, stream);
#endif
  if (dtype != TL_BF16 || Cin != 32 || Cout != 32 || x_ld % 8 || g_ld % 8 || ((uintptr_t)x) % 16 || ((uintptr_t)gout) % 16 || ((uintptr_t)blk_unit) % 16) return TL_ERR_UNSUPPORTED;
  if (n * x_ld * 2 > 0x7FFFFFFFll || n * g_ld * 2 > 0x7FFFFFFFll || n * 128 > 0x7FFFFFFFll) return TL_ERR_UNSUPPORTED;     // 32-bit buffer offsets
  hipStream_t s = tl_s(stream);
  static TlAttrOnce attr_once;
  if (!tl_lds_attr(attr_once, reinterpret_cast<const void*>(&k_wgrad_blk), 160 * 1024)) return TL_ERR_LAUNCH;
  k_wgrad_blk<<<kWB_PARTS, kWB_NT, 2 * kWB_BUF, s>>>(static_cast<const uint16_t*>(x), x_ld, static_cast<const uint16_t*>(gout), g_ld, blk_unit, blk_counter, blk_halo,
                                                      blk_lrb, n, ws);
  if (hipGetLastError() != hipSuccess) return TL_ERR_LAUNCH;
  return tl_launch_wgrad_reduce(ws, kWB_PARTS, 27 * 1024, gw, s, 27, 32, 32, ref_layout);
}

}  // extern "C"
