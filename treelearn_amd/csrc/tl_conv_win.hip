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
#include <atomic>
#include <type_traits>
#include <atomic>

namespace {

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

// NB = Cout / 32, SP = Cin / 32 channel slices, WIN window rows (multiple of 16), RB row blocks of 32 per wave, W waves,
// WR window ring size (window DMA runs WR - 1 steps ahead), CT: row indices come from the column-form rulebook (p.ctab: one base
// per (dx, dy) group + presence mask, 8 B per row and step instead of 12).
// ABL (dev builds only; results are wrong on purpose): 1 no window DMA, 2 no MFMA, 4 no LDS fragment reads, 8 no output stores,
// 16 no weight DMA, 32 no index loads (identity rows), 64 no step barrier, 128 no index decode / window-relative address arithmetic /
// minima exchange (fragments read at a lane-constant offset, window base = tile start): everything else as in the full kernel
template <int NB, int SP, int WIN, int RB, int W, int WR, bool CT, int ABL = 0>
__global__ void __launch_bounds__(W * 64, 2) k_conv_win(ConvP p, int ntiles) {
  constexpr int COUT = NB * 32, CIN = SP * 32, RBYT = 64;
  constexpr int TILE = W * RB * 32, NV = 9 * SP, PD = WR - 1;
  constexpr int WCH = WIN / 16, BCH = 3 * COUT / 16;      // 1-KB DMA chunks (16 rows of 64 B) per step: window, weights
  constexpr int NWD = (WCH + W - 1) / W, NBD = (BCH + W - 1) / W;   // DMA instructions per wave and step
  constexpr int WBYTES = WIN * RBYT, BBYTES = 3 * COUT * RBYT;
  constexpr int BOFF = WR * WBYTES;                       // weight ring (2) behind the window ring (WR)
  constexpr int ZOFF = BOFF + 2 * BBYTES;                 // the all-zero row
  constexpr int LOFF = ZOFF + RBYT;                       // int lox[WR][8]
  constexpr int EP = 36;
  constexpr int NR = CT ? 2 : 3;                          // index words per row and step
  static_assert(W * 32 * EP * 4 <= WBYTES, "epilogue scratch aliases one window buffer");
  static_assert(WIN % 16 == 0 && (3 * COUT) % 16 == 0 && W <= 8 && (WR == 2 || WR == 3) && NV > WR, "configuration");
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
  if (tid < WR * 8) lox[tid] = 0x7FFFFFFF;                                   // slots of waves that do not exist (W < 8)

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
  // raw index words of step (tile ordinal i, v): branch-free (rows past the end read the last row's entry; masked at decode)
  auto load_raw = [&](int64_t t0, int v, int (&dst)[NR][RB]) __attribute__((always_inline)) {   // t0 = first row of the step's tile
    const int g = v / SP;
    const int64_t r0 = t0 + wv * (RB * 32) + fi;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int64_t row = min(r0 + rb * 32, p.n_out - 1);
      if constexpr (ABL & 32) {
#pragma unroll
        for (int t = 0; t < NR; ++t) dst[t][rb] = CT ? (t == 0 ? (int)row + g - 4 : 0x7FFFFFF) : (int)row + g - 4;
      } else if constexpr (CT) {
        dst[0][rb] = p.ctab[(int64_t)g * p.n_out + row];
        dst[1][rb] = p.ctab[(int64_t)9 * p.n_out + row];
      } else {
#pragma unroll
        for (int t = 0; t < 3; ++t) dst[t][rb] = p.table[(int64_t)(3 * g + t) * p.n_out + row];
      }
    }
  };
  // per row block of a step: rel0 = (first present index of the group) - lo, presence bits m3 (bit t = tap t present) and the tap
  // offsets c1, c2 (column form: tap t = base + number of present taps below it; table form: arbitrary indices)
  struct Taps { int idx[3]; unsigned m3; };
  auto decode = [&](const int (&raw)[NR][RB], int64_t t0, int v, int rb) __attribute__((always_inline)) {
    const bool valid = t0 + wv * (RB * 32) + fi + rb * 32 < p.n_out;
    Taps tp;
    if constexpr (CT) {
      tp.m3 = valid ? ((unsigned)raw[1][rb] >> (3 * (v / SP))) & 7u : 0u;
      const int c1 = (int)(tp.m3 & 1u);
      tp.idx[0] = raw[0][rb]; tp.idx[1] = raw[0][rb] + c1; tp.idx[2] = raw[0][rb] + c1 + (int)((tp.m3 >> 1) & 1u);
    } else {
      tp.m3 = 0u;
#pragma unroll
      for (int t = 0; t < 3; ++t) { tp.idx[t] = raw[t][rb]; tp.m3 |= (valid && raw[t][rb] >= 0) ? (1u << t) : 0u; }
    }
    return tp;
  };
  auto min_raw = [&](const int (&raw)[NR][RB], int64_t t0, int v) __attribute__((always_inline)) {
    int mn = 0x7FFFFFFF;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      if constexpr (CT) {                                                                  // the base IS the smallest present index of the group
        const bool valid = t0 + wv * (RB * 32) + fi + rb * 32 < p.n_out;
        const unsigned m3 = ((unsigned)raw[1][rb] >> (3 * (v / SP))) & 7u;
        mn = min(mn, (valid && m3) ? raw[0][rb] : 0x7FFFFFFF);
      } else {
        const Taps tp = decode(raw, t0, v, rb);
#pragma unroll
        for (int t = 0; t < 3; ++t) mn = min(mn, ((tp.m3 >> t) & 1u) ? tp.idx[t] : 0x7FFFFFFF);
      }
    }
    return wave_min(mn);
  };
  auto read_lo = [&](int slot) __attribute__((always_inline)) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(lox + slot * 8);
    int v = min(min((int)a[0], (int)a[1]), min((int)a[2], (int)a[3]));
    if constexpr (W > 4) {
      const u32x4 c = *reinterpret_cast<const u32x4*>(lox + slot * 8 + 4);
      v = min(v, min(min((int)c[0], (int)c[1]), min((int)c[2], (int)c[3])));
    }
    v = __builtin_amdgcn_readfirstlane(v);
    return v == 0x7FFFFFFF ? 0 : v;
  };
  // whole offsets go in the VGPR operand: the bounds check (rows past the end of the input -> zeros) does not cover an SGPR offset.
  // Every wave issues the same number of DMA instructions (chunk index modulo the chunk count: a few chunks are fetched twice).
  auto dma_window = [&](int v, int lo, int ring) __attribute__((always_inline)) {
    const unsigned s_in = (unsigned)lo * (unsigned)in_ld_b + (unsigned)((v % SP) * RBYT) + dma_in;
    if constexpr (ABL & 1) return;
#pragma unroll
    for (int q0 = 0; q0 < NWD; ++q0) {
      const int q = (q0 * W + wv) % WCH;                                                   // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_ptr)(smem + ring * WBYTES + q * 1024), 16, (int)(s_in + (unsigned)(q * 16) * (unsigned)in_ld_b), 0, 0, 0);
    }
  };
  auto dma_weights = [&](int v, int ring) __attribute__((always_inline)) {
    const unsigned s_w = (unsigned)((3 * (v / SP) * COUT) * (CIN * 2) + (v % SP) * RBYT) + dma_w;
    if constexpr (ABL & 16) return;
#pragma unroll
    for (int q0 = 0; q0 < NBD; ++q0) {
      const int c = (q0 * W + wv) % BCH;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + BOFF + ring * BBYTES + c * 1024), 16, (int)(s_w + (unsigned)(c * 16 * (CIN * 2))), 0, 0, 0);
    }
  };

  // ---- prologue (NV > WR: the first WR steps are in the first tile): raw indices and window bases of steps 0 .. WR-1, windows
  // 0 .. PD-1, weights 0
  int raw[WR + 1][NR][RB];                                                                // [k] = step s + k
  int lo_q[WR];                                                                           // window bases of steps s .. s + PD
  int64_t t_cur = (int64_t)tile_of(0) * TILE, t_pf = t_cur;                              // first rows of the current tile and of the tile of the step WR ahead
#pragma unroll
  for (int k = 0; k < WR; ++k) load_raw(t_cur, k, raw[k]);
  __syncthreads();                                                                        // lox / zero-row initialisation
#pragma unroll
  for (int k = 0; k < WR; ++k) {
    const int mn = min_raw(raw[k], t_cur, k);
    if (lane == 0) lox[k * 8 + wv] = mn;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PD; ++k) { lo_q[k] = read_lo(k); dma_window(k, lo_q[k], k); }
  dma_weights(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][nb][i] = 0.f;

  // coordinates of the steps 1 .. WR ahead (clamped to this workgroup's last step); ring positions of the current step
  int ia[WR + 1], va[WR + 1];
#pragma unroll
  for (int k = 0; k <= WR; ++k) { ia[k] = 0; va[k] = k; }
  int wr = 0, br = 0;

  auto step = [&](int v, auto last_c) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
    const int64_t r0 = t_cur + wv * (RB * 32);
    // 1. requests, oldest first: this tile's residual (tile ends here), indices WR steps ahead, weights one step ahead, window
    //    PD steps ahead
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
    if constexpr ((ABL & 128) == 0) load_raw(t_pf, va[WR], raw[WR]);
    const int wpd = wr + PD >= WR ? wr + PD - WR : wr + PD;                                // (s + PD) % WR
    if constexpr (ABL & 128) lo_q[PD] = (int)min(t_pf, p.n_in - 1);
    else lo_q[PD] = read_lo(wpd);
    dma_weights(va[1], br ^ 1);
    dma_window(va[PD], lo_q[PD], wpd);
    __builtin_amdgcn_sched_barrier(0);

    // 2. this step: three taps from the staged window
    {
      const int sp = v % SP;
      const char* bb = smem + BOFF + br * BBYTES;
      const int wbase = wr * WBYTES;
      Taps tp[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) tp[rb] = decode(raw[0], t_cur, v, rb);
      const int swz = fh * 16 + wbase;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        u32x4 A[RB][2], B[NB][2];
        bool anyout = false;
        bool outl[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const int rel = tp[rb].idx[t] - lo_q[0];
          const bool present = (tp[rb].m3 >> t) & 1u;
          const bool inwin = (unsigned)rel < (unsigned)WIN;
          outl[rb] = present && !inwin;
          anyout |= outl[rb];
          // logical piece 2 j + fh of window row rel sits at physical piece (2 j + fh) ^ ((rel >> 2) & 3): j = 1 is j = 0 with bit 5 flipped
          int off0 = (present && inwin) ? ((rel * RBYT + swz) ^ (((rel >> 2) & 3) * 16)) : ZOFF;
          if constexpr (ABL & 128) off0 = ((fi + 32 * rb + 8 * t) * RBYT + swz) ^ (((fi >> 2) & 3) * 16);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int off = off0 ^ (j * 32);
            if constexpr (ABL & 4) A[rb][j] = u32x4{(unsigned)off, (unsigned)off, (unsigned)off, (unsigned)off};
            else A[rb][j] = *reinterpret_cast<const u32x4*>(smem + off);
          }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if constexpr (ABL & 4) B[nb][j] = u32x4{(unsigned)bsw[j], (unsigned)t, (unsigned)nb, (unsigned)br};
            else B[nb][j] = *reinterpret_cast<const u32x4*>(bb + (t * COUT + nb * 32) * RBYT + bsw[j]);
          }
        if (__builtin_amdgcn_ballot_w64(anyout) != 0) {                                   // rare: present neighbour outside the window
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const unsigned off = outl[rb] ? (unsigned)tp[rb].idx[t] * (unsigned)in_ld_b + (unsigned)(sp * RBYT + (2 * j + fh) * 16) : 0xFFFFFFF0u;
              const u32x4 gq = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (int)off, 0, 0));
              A[rb][j] |= gq;                                                              // in-window / absent lanes got zeros from the bounds check
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
              if constexpr (ABL & 2) acc[rb][nb][(t + j) & 15] += __uint_as_float(A[rb][j][0] ^ B[nb][j][1]);
              else mma16<true>(acc[rb][nb], A[rb][j], B[nb][j]);
            }
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // 3. partial minimum of the indices WR steps ahead into the ring slot the current step's base came from; then everything but
    //    the youngest window request has landed (vmcnt retires in order; WR == 2: that request is the next step's window); barrier
    if constexpr ((ABL & 128) == 0) {
      const int mn = min_raw(raw[WR], t_pf, va[WR]);
      if (lane == 0) lox[wr * 8 + wv] = mn;
    }
    constexpr int KEEP = (WR == 2 || (ABL & 1)) ? 0 : NWD;
    if constexpr (ABL & 64) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(KEEP) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(KEEP) : "memory");

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
            if (orow < p.n_out && !((ABL & 8) && acc[rb][nb][0] != 1.2345e30f)) {
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
    for (int k = 0; k < WR; ++k)
#pragma unroll
      for (int t = 0; t < NR; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) raw[k][t][rb] = raw[k + 1][t][rb];
#pragma unroll
    for (int k = 0; k < PD; ++k) lo_q[k] = lo_q[k + 1];
#pragma unroll
    for (int k = 0; k < WR; ++k) { ia[k] = ia[k + 1]; va[k] = va[k + 1]; }
    if (++va[WR] == NV) {                                                                  // the step WR ahead enters the next tile
      va[WR] = 0; ++ia[WR];
      if (ia[WR] >= m) { ia[WR] = m - 1; va[WR] = NV - 1; }
      else t_pf = (int64_t)tile_of(ia[WR]) * TILE;
    }
    wr = wr == WR - 1 ? 0 : wr + 1; br ^= 1;
  };

  for (int i = 0; i < m; ++i) {
    t_cur = (int64_t)tile_of(i) * TILE;
    for (int v = 0; v < NV - 1; ++v) step(v, std::false_type{});
    step(NV - 1, std::true_type{});
  }
}

template <int NB, int SP, int WIN, int RB, int W, int WR, int ABL = 0>
int launch(ConvP p, hipStream_t s) {
  constexpr int TILE = W * RB * 32;
  const size_t lds = (size_t)WR * WIN * 64 + 2 * (size_t)3 * NB * 32 * 64 + 64 + (size_t)WR * 32;
  constexpr int per_cu = W == 8 ? 1 : 2;                    // workgroups resident per CU (LDS: 160 KB, 16 waves)
  if (lds * per_cu > 160 * 1024) return TL_ERR_UNSUPPORTED;
  const int ntiles = (int)tl_cdiv(p.n_out, TILE);
  const int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;   // persistent; a multiple of 8 keeps tile ordinal i of block b on XCD b % 8
  const bool ct = p.ctab != nullptr;
  auto go = [&](auto kern) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return (int)TL_ERR_LAUNCH;
      attr_set = true;
    }
    kern<<<grid, W * 64, lds, s>>>(p, ntiles);
    return hipGetLastError() == hipSuccess ? (int)TL_OK : (int)TL_ERR_LAUNCH;
  };
  return ct ? go(&k_conv_win<NB, SP, WIN, RB, W, WR, true, ABL>) : go(&k_conv_win<NB, SP, WIN, RB, W, WR, false, ABL>);
}

}  // namespace

int g_win_rows = 0;     // tl_set_tuning("win_rows"): 0 = per shape; 512 = the 8-wave / 512-row / ring-3 form for every shape
int g_win_ct = 1;       // tl_set_tuning("win_ct"): use the column-form rulebook when the caller provides it

#ifdef TL_DEV
static int g_win_abl = 0;
extern "C" int tl_dev_win_abl(int mode) { g_win_abl = mode; return TL_OK; }      // dev build only: ablation variant of the 64 -> 64 shape
#endif

int tl_launch_conv_win(const ConvP& p0, hipStream_t s) {
  ConvP p = p0;
  if (!g_win_ct) p.ctab = nullptr;
  if (p.K != 27 || !p.table || p.in_scale || p.in_relu || p.Cin % 32 || p.Cout % 32 || p.one_hot) return TL_ERR_UNSUPPORTED;
  const int64_t ld_b = p.in_ld * 2, in_bytes = (p.n_in - 1) * ld_b + (int64_t)p.Cin * 2;
  if (!(in_bytes > 0 && in_bytes + 1024 * ld_b < 0xFFFFFFFFll)) return TL_ERR_UNSUPPORTED;
  const int nb = p.Cout / 32, sp = p.Cin / 32;
#ifdef TL_DEV
  if (g_win_abl && nb == 2 && sp == 2) {
    switch (g_win_abl) {
#define TL_A(M_) case M_: return launch<2, 2, 352, 2, 4, 2, M_>(p, s);
      TL_A(1) TL_A(2) TL_A(4) TL_A(8) TL_A(16) TL_A(17) TL_A(32) TL_A(64) TL_A(6) TL_A(23) TL_A(55) TL_A(128) TL_A(136)
#undef TL_A
    }
  }
#endif
  // two forms: 4 waves x 64 rows (256-row tiles, window ring of 2, two workgroups per CU whose phases interleave) and
  // 8 waves x 64 rows (512-row tiles, ring of 3, one workgroup per CU)
#define TL_W(NB_, SP_, WIN4_)                                                                                   \
  if (nb == NB_ && sp == SP_) return g_win_rows == 512 ? launch<NB_, SP_, 640, 2, 8, 3>(p, s) : launch<NB_, SP_, WIN4_, 2, 4, 2>(p, s);
  TL_W(2, 2, 352) TL_W(2, 4, 352) TL_W(3, 3, 320) TL_W(3, 6, 320) TL_W(1, 1, 384) TL_W(1, 2, 384)
#undef TL_W
  return TL_ERR_UNSUPPORTED;
}
