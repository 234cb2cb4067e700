// Sparse SubM conv, "window" form (bf16, 27 taps, Cin % 32 == 0): the three dz taps of a (dx, dy) column share ONE staged copy
// of their input rows.
//
// Why.  In ascending-key order (z fastest) the dz = -1 / 0 / +1 neighbours of a voxel in column (x+dx, y+dy) are consecutive
// rows, and the neighbours of TILE consecutive output rows in one (dx, dy) group lie in a row range barely longer than TILE
// (measured on the config-2 rulebooks, 512-row tiles: median span 513 rows, 99.7 % of the present pairs inside 640 rows).  The
// register-gather kernels (tl_conv_streamq.hip) pull every (row, tap) pair through the vector-memory path on its own: 27 row
// reads per output row at the L1 rate of 64 B/clk/CU, which is exactly what the matrix pipes consume at 64 -> 64 channels --
// gathers and MFMAs co-limit and the kernel sits at a third of either.  Here a workgroup owns TILE = 512 consecutive output
// rows and walks 9 groups x SP channel slices ("steps").  Per step the WINDOW [lo, lo + WIN) of input rows of that group and
// the three taps' weight slices are copied global -> LDS by LDS-DMA (buffer_load ... lds: fully coalesced 64-B row segments, no
// VGPR staging, XOR-swizzled through the SOURCE address), double-buffered one step ahead; A fragments are then read from LDS
// by row index -- 1.25 window rows per output row and group instead of 3 gathered rows, i.e. 11 instead of 27 row reads per
// output row through L1, and the per-lane fragment reads move to the 256 B/clk LDS.  Absent neighbours read an all-zero row;
// the rare present neighbour outside the window (tile straddling a sparse region) is fetched straight from global memory
// (wave-uniform slow path), so the result never depends on the window choice.
//
// lo of a step = min over the tile's present indices of the group, computed in-kernel: every wave reduces the indices it will
// need two steps ahead and the eight partial minima cross through LDS with the step barrier that exists anyway.
//
// Per step and wave: 3 taps x 2 k-steps x NB x RB MFMAs (32x32x16 bf16), 6 + 6 NB/2 ds_read_b128, one barrier.  Per CU and
// step (NB = 2): 125 B/clk of LDS reads, 34 B/clk of DMA fill -- each about half of what the matrix pipes would allow.
// Deterministic (fixed summation order: groups ascending, taps dz ascending, channels ascending).
#include "tl_conv_internal.h"

namespace {

constexpr int kWaves = 8;

static __device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}

// NB = Cout / 32, SP = Cin / 32 channel slices, WIN window rows (multiple of 16), RB row blocks of 32 per wave
template <int NB, int SP, int WIN, int RB>
__global__ void __launch_bounds__(kWaves * 64, 2) k_conv_win(ConvP p, int ntiles) {
  constexpr int W = kWaves, NTH = W * 64;
  constexpr int COUT = NB * 32, CIN = SP * 32, RBYT = 64;
  constexpr int TILE = W * RB * 32, NV = 9 * SP;
  constexpr int WCH = WIN / 16, BCH = 3 * COUT / 16;      // 1-KB DMA chunks (16 rows of 64 B) per step: window, weights
  constexpr int WBYTES = WIN * RBYT, BBYTES = 3 * COUT * RBYT;
  constexpr int ZOFF = 2 * WBYTES + 2 * BBYTES;           // the all-zero row
  constexpr int LOFF = ZOFF + RBYT;                       // int lox[2][8]
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
  const int dma_in = (lane >> 2) * in_ld_b + sw16;
  const int dma_w = (lane >> 2) * (CIN * 2) + sw16;
  // fragment read constants
  const int bsw[2] = {(((0 + fh) ^ ((fi >> 2) & 3)) * 16) + fi * RBYT, (((2 + fh) ^ ((fi >> 2) & 3)) * 16) + fi * RBYT};

  auto tile_of = [&](int i) { return xcd_tile(b + i * G, ntiles); };
  auto load_idx = [&](int i, int v, int (&dst)[3][RB]) __attribute__((always_inline)) {   // indices of step (tile ordinal i, v); -1 beyond the end
    const int g = v / SP;
    const int64_t r0 = (int64_t)tile_of(i) * TILE + wv * (RB * 32) + fi;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int64_t row = r0 + rb * 32;
        dst[t][rb] = row < p.n_out ? p.table[(int64_t)(3 * g + t) * p.n_out + row] : -1;
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
  auto issue_dma = [&](int v, int lo, int buf) __attribute__((always_inline)) {           // window + weights of step v -> buffers `buf`
    const int g = v / SP, sp = v % SP;
    // whole offsets in the VGPR operand: the bounds check (rows past the end of the input -> zeros) does not cover an SGPR offset
    const unsigned s_in = (unsigned)lo * (unsigned)in_ld_b + (unsigned)(sp * RBYT) + (unsigned)dma_in;
    const unsigned s_w = (unsigned)((3 * g * COUT) * (CIN * 2) + sp * RBYT) + (unsigned)dma_w;
#pragma unroll
    for (int q0 = 0; q0 < WCH + BCH; q0 += W) {
      const int q = q0 + wv;                                                              // wave-uniform
      if (q < WCH) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_ptr)(smem + buf * WBYTES + q * 1024), 16, (int)(s_in + (unsigned)(q * 16) * (unsigned)in_ld_b), 0, 0, 0);
      } else if (q < WCH + BCH) {
        const int c = q - WCH;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + 2 * WBYTES + buf * BBYTES + c * 1024), 16, (int)(s_w + (unsigned)(c * 16 * (CIN * 2))), 0, 0, 0);
      }
    }
  };

  // ---- prologue: indices of steps 0 and 1, window base of step 0, first DMA
  int ic[3][RB], in1[3][RB], in2[3][RB];
  load_idx(0, 0, ic);
  load_idx(0, 1, in1);                                                                    // NV >= 9: step 1 is in the same tile
  {
    const int m0 = min_idx(ic), m1 = min_idx(in1);
    if (lane == 0) { lox[wv] = m0; lox[8 + wv] = m1; }
  }
  __syncthreads();
  int lo_c = read_lo(0);
  issue_dma(0, lo_c, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][nb][i] = 0.f;

  // step counters: current (i, v), next (i1, v1), next-but-one (i2, v2)
  int i = 0, v = 0;
  int i1 = 0, v1 = 1, i2 = 0, v2 = 2;
  int cur = 0;
  u32x4 resv[RB][NB][2];

  while (i < m) {
    const bool last = (v == NV - 1);
    // 1. window base + DMA of the next step, indices of the step after it, this tile's residual if the tile ends here
    int lo_n = 0;
    if (i1 < m) {
      lo_n = read_lo(cur ^ 1);
      issue_dma(v1, lo_n, cur ^ 1);
    }
    if (i2 < m) load_idx(i2, v2, in2);
    else {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) in2[t][rb] = -1;
    }
    const int64_t r0 = (int64_t)tile_of(i) * TILE + wv * (RB * 32);
    if (last && p.res) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int e0 = 0; e0 < 2; ++e0) {
            const int e = lane + e0 * 64, rr = e >> 2, cvv = e & 3;
            const int64_t orow = r0 + rb * 32 + rr;
            resv[rb][nb][e0] = orow < p.n_out ? *reinterpret_cast<const u32x4*>((const uint16_t*)p.res + orow * p.res_ld + nb * 32 + cvv * 8) : u32x4{0u, 0u, 0u, 0u};
          }
    }
    __builtin_amdgcn_sched_barrier(0);

    // 2. this step: three taps from the staged window
    {
      const int sp = v % SP;
      const char* bb = smem + 2 * WBYTES + cur * BBYTES;
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
          const int base = rel * RBYT;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int off = inwin ? base + (((2 * j + fh) ^ sz) * 16) + cur * WBYTES : ZOFF;
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

    // 3. partial minimum of the indices two steps ahead; everything this step requested has landed; step barrier
    {
      const int mn = min_idx(in2);
      if (lane == 0) lox[cur * 8 + wv] = mn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // 4. tile finished: epilogue through the window buffer this step read (free now), then one more barrier before it is refilled
    if (last) {
      float* ew = reinterpret_cast<float*>(smem + cur * WBYTES) + wv * 32 * EP;
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
      __syncthreads();
    }

    // 5. rotate
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) { ic[t][rb] = in1[t][rb]; in1[t][rb] = in2[t][rb]; }
    lo_c = lo_n;
    i = i1; v = v1; i1 = i2; v1 = v2;
    if (++v2 == NV) { v2 = 0; ++i2; }
    cur ^= 1;
  }
}

template <int NB, int SP, int WIN, int RB>
int launch(ConvP p, hipStream_t s) {
  constexpr int TILE = kWaves * RB * 32;
  const size_t lds = 2 * (size_t)WIN * 64 + 2 * (size_t)3 * NB * 32 * 64 + 64 + 64;
  if (lds > 160 * 1024) return TL_ERR_UNSUPPORTED;
  static bool attr_set = false;
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

int g_win_rows = 640;   // tl_set_tuning("win_rows"): 640 or 768

int tl_launch_conv_win(const ConvP& p, hipStream_t s) {
  if (p.K != 27 || !p.table || p.in_scale || p.in_relu || p.Cin % 32 || p.Cout % 32 || p.one_hot) return TL_ERR_UNSUPPORTED;
  const int64_t ld_b = p.in_ld * 2, in_bytes = (p.n_in - 1) * ld_b + (int64_t)p.Cin * 2;
  if (!(in_bytes > 0 && in_bytes + 1024 * ld_b < 0xFFFFFFFFll)) return TL_ERR_UNSUPPORTED;
  const int nb = p.Cout / 32, sp = p.Cin / 32;
#define TL_W(NB_, SP_)                                                                       \
  if (nb == NB_ && sp == SP_) return g_win_rows >= 768 ? launch<NB_, SP_, 768, 2>(p, s) : launch<NB_, SP_, 640, 2>(p, s);
  TL_W(2, 2) TL_W(2, 4) TL_W(3, 3) TL_W(3, 6) TL_W(4, 4) TL_W(4, 8) TL_W(1, 1) TL_W(1, 2)
#undef TL_W
  return TL_ERR_UNSUPPORTED;
}
