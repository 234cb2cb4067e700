// Sparse SubM conv, "window" form (bf16, 27 taps, Cin % 32 == 0): the three dz taps of a (dx, dy) column share ONE staged copy
// of their input rows.
//
// Why.  In ascending-key order (z fastest) the dz = -1 / 0 / +1 neighbours of a voxel in column (x+dx, y+dy) are consecutive
// rows, and the neighbours of TILE consecutive output rows in one (dx, dy) group lie in a row range barely longer than TILE
// (measured on the config-2 rulebooks, 512-row tiles: median span 513 rows, 99.7 % of the present pairs inside 640 rows).  The
// register-gather kernels (tl_conv_streamq.hip) pull every (row, tap) pair through the vector-memory path on its own: 27 row
// reads per output row at the L1 rate of 64 B/clk/CU, which is exactly what the matrix pipes consume at 64 -> 64 channels --
// gathers and MFMAs co-limit and the kernel sits at a third of either.  Here a workgroup owns TILE = 512 consecutive output
// rows and walks 9 groups x SP channel slices of 32 ("steps").  Per step the WINDOW [lo, lo + WIN) of input rows of that group
// and the three taps' weight slices are copied global -> LDS by LDS-DMA (buffer_load ... lds: fully coalesced 64-B row segments,
// no VGPR staging, XOR-swizzled through the SOURCE address); A fragments are then read from LDS by row index -- 1.25 window
// rows per output row and group instead of 3 gathered rows, i.e. 11 instead of 27 row reads per output row through L1, and the
// per-lane fragment reads move to the 256 B/clk LDS.  Absent neighbours read an all-zero row; the rare present neighbour outside
// the window (tile straddling a sparse region) is fetched straight from global memory (wave-uniform slow path), so the result
// never depends on the window choice.
//
// Pipeline (one step = 768 MFMA clocks per wave, shorter than a loaded-chip DMA round trip, so the window runs TWO steps ahead):
//   window ring of 3, weight ring of 2, row indices loaded three steps ahead, window bases two steps ahead.
//   step s:  request indices(s+3) | lo(s+2) <- minima exchanged through LDS | DMA weights(s+1), window(s+2) |
//            3 taps x 2 k-steps x NB x RB MFMAs from window(s), weights(s) | partial minima of indices(s+3) -> LDS |
//            s_waitcnt vmcnt(#window DMAs): everything but window(s+2) has landed | s_barrier
//   lo of a step = min over the tile's present indices of the group: every wave reduces its rows with DPP, the eight partial
//   minima cross through LDS with the step barrier that exists anyway.
// The loop body is branch-free (steps past the end re-request the last step's data), so hipcc's vmcnt accounting stays exact.
// Per CU and step (NB = 2): 125 B/clk of LDS fragment reads, 34 B/clk of DMA fill.
// Deterministic (fixed summation order: groups ascending, taps dz ascending, channels ascending).
#include "tl_conv_internal.h"
#include <type_traits>
#include <atomic>

namespace {

constexpr int kWaves = 8;

template <int CTRL>
static __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }

// minimum over the wave without LDS round trips: quad butterflies, row mirrors, then the four rows through SGPRs
static __device__ __forceinline__ int wave_min(int v) {
  v = min(v, dpp_i<0xB1>(v));        // quad_perm [1,0,3,2]
  v = min(v, dpp_i<0x4E>(v));        // quad_perm [2,3,0,1]
  v = min(v, dpp_i<0x141>(v));       // row_half_mirror
  v = min(v, dpp_i<0x140>(v));       // row_mirror
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// NB = Cout / 32, SP = Cin / 32 channel slices, WIN window rows (multiple of 16), RB row blocks of 32 per wave
template <int NB, int SP, int WIN, int RB>
__global__ void __launch_bounds__(kWaves * 64, 2) k_conv_win(ConvP p, int ntiles) {
  constexpr int W = kWaves;
  constexpr int COUT = NB * 32, CIN = SP * 32, RBYT = 64;
  constexpr int TILE = W * RB * 32, NV = 9 * SP;
  constexpr int WCH = WIN / 16, BCH = 3 * COUT / 16;      // 1-KB DMA chunks (16 rows of 64 B) per step: window, weights
  constexpr int NWD = (WCH + W - 1) / W, NBD = (BCH + W - 1) / W;   // DMA instructions per wave and step
  constexpr int WBYTES = WIN * RBYT, BBYTES = 3 * COUT * RBYT;
  constexpr int BOFF = 3 * WBYTES;                        // weight ring behind the window ring
  constexpr int ZOFF = BOFF + 2 * BBYTES;                 // the all-zero row
  constexpr int LOFF = ZOFF + RBYT;                       // int lox[3][8]
  constexpr int EP = 36;
  static_assert(W * 32 * EP * 4 <= WBYTES, "epilogue scratch aliases one window buffer");
  static_assert(WIN % 16 == 0 && (3 * COUT) % 16 == 0, "whole DMA chunks");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, fh = lane >> 5;
  const int G = (int)gridDim.x;
  const int b = (int)blockIdx.x;
  const int m = b < ntiles ? (ntiles - 1 - b) / G + 1 : 0;                   // tiles of this workgroup: xcd_tile(b + i G), i < m
  if (m == 0) return;

  if (tid < RBYT / 4) reinterpret_cast<int*>(smem + ZOFF)[tid] = 0;
  int* lox = reinterpret_cast<int*>(smem + LOFF);

  const int in_ld_b = (int)(p.in_ld * 2);
  const int64_t in_bytes = ((int64_t)p.n_in - 1) * in_ld_b + (int64_t)CIN * 2;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, 27 * COUT * CIN * 2, 0x00020000);
  // DMA lane constants: lane L of a chunk fills LDS slot L (16 B): row L >> 2 of the chunk, physical piece L & 3, which must hold
  // logical piece (L & 3) ^ ((row >> 2) & 3) -- the involution the fragment reads apply
  const int sw16 = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
  const unsigned dma_in = (unsigned)((lane >> 2) * in_ld_b + sw16);
  const unsigned dma_w = (unsigned)((lane >> 2) * (CIN * 2) + sw16);
  // fragment read constants
  const int bsw[2] = {(((0 + fh) ^ ((fi >> 2) & 3)) * 16) + fi * RBYT, (((2 + fh) ^ ((fi >> 2) & 3)) * 16) + fi * RBYT};

  auto tile_of = [&](int i) { return xcd_tile(b + i * G, ntiles); };
  // indices of step (tile ordinal i, v): branch-free (rows past the end read the last row's entry and are masked afterwards)
  auto load_idx = [&](int i, int v, int (&dst)[3][RB]) __attribute__((always_inline)) {
    const int g = v / SP;
    const int64_t r0 = (int64_t)tile_of(i) * TILE + wv * (RB * 32) + fi;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int64_t row = r0 + rb * 32;
        const int val = p.table[(int64_t)(3 * g + t) * p.n_out + min(row, p.n_out - 1)];
        dst[t][rb] = row < p.n_out ? val : -1;
      }
  };
  auto min_idx = [&](const int (&ix)[3][RB]) __attribute__((always_inline)) {
    int v = 0x7FFFFFFF;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) v = min(v, ix[t][rb] >= 0 ? ix[t][rb] : 0x7FFFFFFF);
    return wave_min(v);
  };
  auto read_lo = [&](int slot) __attribute__((always_inline)) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(lox + slot * 8), c = *reinterpret_cast<const u32x4*>(lox + slot * 8 + 4);
    int v = min(min(min((int)a[0], (int)a[1]), min((int)a[2], (int)a[3])), min(min((int)c[0], (int)c[1]), min((int)c[2], (int)c[3])));
    v = __builtin_amdgcn_readfirstlane(v);
    return v == 0x7FFFFFFF ? 0 : v;
  };
  // whole offsets go in the VGPR operand: the bounds check (rows past the end of the input -> zeros) does not cover an SGPR offset.
  // Every wave issues the same number of DMA instructions (chunk index modulo the chunk count: a few chunks are fetched twice).
  auto dma_window = [&](int v, int lo, int ring) __attribute__((always_inline)) {
    const unsigned s_in = (unsigned)lo * (unsigned)in_ld_b + (unsigned)((v % SP) * RBYT) + dma_in;
#pragma unroll
    for (int q0 = 0; q0 < NWD; ++q0) {
      const int q = (q0 * W + wv) % WCH;                                                   // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_ptr)(smem + ring * WBYTES + q * 1024), 16, (int)(s_in + (unsigned)(q * 16) * (unsigned)in_ld_b), 0, 0, 0);
    }
  };
  auto dma_weights = [&](int v, int ring) __attribute__((always_inline)) {
    const unsigned s_w = (unsigned)((3 * (v / SP) * COUT) * (CIN * 2) + (v % SP) * RBYT) + dma_w;
#pragma unroll
    for (int q0 = 0; q0 < NBD; ++q0) {
      const int c = (q0 * W + wv) % BCH;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + BOFF + ring * BBYTES + c * 1024), 16, (int)(s_w + (unsigned)(c * 16 * (CIN * 2))), 0, 0, 0);
    }
  };

  // ---- prologue: indices of steps 0..2, window bases of steps 0 and 1, windows 0 and 1, weights 0  (NV >= 9: same tile)
  int ic[3][RB], in1[3][RB], in2[3][RB], in3[3][RB];
  load_idx(0, 0, ic); load_idx(0, 1, in1); load_idx(0, 2, in2);
  {
    const int m0 = min_idx(ic), m1 = min_idx(in1), m2 = min_idx(in2);
    if (lane == 0) { lox[wv] = m0; lox[8 + wv] = m1; lox[16 + wv] = m2; }
  }
  __syncthreads();
  int lo_c = read_lo(0), lo_1 = read_lo(1);
  dma_window(0, lo_c, 0); dma_weights(0, 0); dma_window(1, lo_1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][nb][i] = 0.f;

  // coordinates of the steps one, two and three ahead (clamped to this workgroup's last step)
  int v1 = 1, i2 = 0, v2 = 2, i3 = 0, v3 = 3;
  int wr = 0, br = 0;                                                                     // ring positions of the current step
  auto advance = [&](int& ii, int& vv) __attribute__((always_inline)) {
    if (++vv == NV) { vv = 0; ++ii; }
    if (ii >= m) { ii = m - 1; vv = NV - 1; }
  };

  auto step = [&](int i, int v, auto last_c) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
    const int64_t r0 = (int64_t)tile_of(i) * TILE + wv * (RB * 32);
    // 1. requests, oldest first: this tile's residual (tile ends here), indices three steps ahead, weights one step ahead,
    //    window two steps ahead
    [[maybe_unused]] u32x4 resv[RB][NB][2];
    if constexpr (LAST) {
      if (p.res) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e0 = 0; e0 < 2; ++e0) {
              const int e = lane + e0 * 64, rr = e >> 2, cvv = e & 3;
              const int64_t orow = min(r0 + rb * 32 + rr, p.n_out - 1);
              resv[rb][nb][e0] = *reinterpret_cast<const u32x4*>((const uint16_t*)p.res + orow * p.res_ld + nb * 32 + cvv * 8);
            }
      }
    }
    load_idx(i3, v3, in3);
    const int wr2 = wr >= 1 ? wr - 1 : 2;                                                  // (s + 2) % 3
    const int lo_2 = read_lo(wr2);
    dma_weights(v1, br ^ 1);
    dma_window(v2, lo_2, wr2);
    __builtin_amdgcn_sched_barrier(0);

    // 2. this step: three taps from the staged window
    {
      const int sp = v % SP;
      const char* bb = smem + BOFF + br * BBYTES;
      const int wbase = wr * WBYTES;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        u32x4 A[RB][2], B[NB][2];
        bool anyout = false;
        bool outl[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const int idx = ic[t][rb];
          const int rel = idx - lo_c;
          const bool inwin = (unsigned)rel < (unsigned)WIN;
          outl[rb] = idx >= 0 && !inwin;
          anyout |= outl[rb];
          const int sz = (rel >> 2) & 3;
          const int base = rel * RBYT + wbase;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int off = inwin ? base + (((2 * j + fh) ^ sz) * 16) : ZOFF;
            A[rb][j] = *reinterpret_cast<const u32x4*>(smem + off);
          }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int j = 0; j < 2; ++j) B[nb][j] = *reinterpret_cast<const u32x4*>(bb + (t * COUT + nb * 32) * RBYT + bsw[j]);
        if (__builtin_amdgcn_ballot_w64(anyout) != 0) {                                   // rare: present neighbour outside the window
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const unsigned off = outl[rb] ? (unsigned)ic[t][rb] * (unsigned)in_ld_b + (unsigned)(sp * RBYT + (2 * j + fh) * 16) : 0xFFFFFFF0u;
              const u32x4 gq = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (int)off, 0, 0));
              A[rb][j] |= gq;                                                              // in-window / absent lanes got zeros from the bounds check
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) mma16<true>(acc[rb][nb], A[rb][j], B[nb][j]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // 3. partial minimum of the indices three steps ahead into the slot this step's base came from two steps ago; then
    //    everything but the window requested in this step has landed (vmcnt retires in order); step barrier
    {
      const int mn = min_idx(in3);
      if (lane == 0) lox[wr * 8 + wv] = mn;
    }
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(NWD) : "memory");

    // 4. tile finished: epilogue through the window buffer this step read (free now), then one more barrier before it is refilled
    if constexpr (LAST) {
      float* ew = reinterpret_cast<float*>(smem + wr * WBYTES) + wv * 32 * EP;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) ew[((r & 3) + 8 * (r >> 2) + 4 * fh) * EP + fi] = acc[rb][nb][r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int e0 = 0; e0 < 2; ++e0) {
            const int e = lane + e0 * 64, rr = e >> 2, cvv = e & 3;
            const int64_t orow = r0 + rb * 32 + rr;
            if (orow < p.n_out) {
              const f32x4 v0 = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8), v1q = *reinterpret_cast<const f32x4*>(ew + rr * EP + cvv * 8 + 4);
              float y[8] = {v0[0], v0[1], v0[2], v0[3], v1q[0], v1q[1], v1q[2], v1q[3]};
              if (p.res) {
                const u32x4 rv = resv[rb][nb][e0];
#pragma unroll
                for (int q = 0; q < 4; ++q) { y[2 * q] += bf16_lo(rv[q]); y[2 * q + 1] += bf16_hi(rv[q]); }
              }
              const int c0 = nb * 32 + cvv * 8;
              epi_store8<true>(p.out, p.out_ld, p.out_scale, p.out_shift, p.out_relu, orow, c0, y);
              if (p.out2) epi_store8<true>(p.out2, p.out2_ld, p.out2_scale, p.out2_shift, p.out2_relu, orow, c0, y);
              if (p.out3) epi_store8<true>(p.out3, p.out3_ld, p.out3_scale, p.out3_shift, p.out3_relu, orow, c0, y);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rb][nb][r] = 0.f;
        }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // 5. rotate
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) { ic[t][rb] = in1[t][rb]; in1[t][rb] = in2[t][rb]; in2[t][rb] = in3[t][rb]; }
    lo_c = lo_1; lo_1 = lo_2;
    v1 = v2; v2 = v3; i2 = i3;
    advance(i3, v3);
    wr = wr == 2 ? 0 : wr + 1; br ^= 1;
  };

  for (int i = 0; i < m; ++i) {
    for (int v = 0; v < NV - 1; ++v) step(i, v, std::false_type{});
    step(i, NV - 1, std::true_type{});
  }
}

template <int NB, int SP, int WIN, int RB>
int launch(ConvP p, hipStream_t s) {
  constexpr int TILE = kWaves * RB * 32;
  const size_t lds = 3 * (size_t)WIN * 64 + 2 * (size_t)3 * NB * 32 * 64 + 64 + 96;
  if (lds > 160 * 1024) return TL_ERR_UNSUPPORTED;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_win<NB, SP, WIN, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return TL_ERR_LAUNCH;
    attr_set = true;
  }
  const int ntiles = (int)tl_cdiv(p.n_out, TILE);
  const int grid = ntiles < 256 ? ntiles : 256;             // persistent: one workgroup per CU; 256 % 8 == 0 keeps tile ordinal i of block b on XCD b % 8
  k_conv_win<NB, SP, WIN, RB><<<grid, kWaves * 64, lds, s>>>(p, ntiles);
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

}  // namespace

int g_win_rows = 640;   // tl_set_tuning("win_rows"): 640 (default) or 576; 128-channel outputs always take 576 (LDS)

int tl_launch_conv_win(const ConvP& p, hipStream_t s) {
  if (p.K != 27 || !p.table || p.in_scale || p.in_relu || p.Cin % 32 || p.Cout % 32 || p.one_hot) return TL_ERR_UNSUPPORTED;
  const int64_t ld_b = p.in_ld * 2, in_bytes = (p.n_in - 1) * ld_b + (int64_t)p.Cin * 2;
  if (!(in_bytes > 0 && in_bytes + 1024 * ld_b < 0xFFFFFFFFll)) return TL_ERR_UNSUPPORTED;
  const int nb = p.Cout / 32, sp = p.Cin / 32;
#define TL_W(NB_, SP_)                                                                       \
  if (nb == NB_ && sp == SP_) return g_win_rows < 640 ? launch<NB_, SP_, 576, 2>(p, s) : launch<NB_, SP_, 640, 2>(p, s);
  TL_W(2, 2) TL_W(2, 4) TL_W(3, 3) TL_W(3, 6) TL_W(1, 1) TL_W(1, 2)      // 128 output channels (NB = 4) spill at 2 waves per SIMD
#undef TL_W
  return TL_ERR_UNSUPPORTED;
}
