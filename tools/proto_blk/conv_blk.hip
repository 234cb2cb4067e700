// PROTOTYPE (not part of the package; built on the GPU box by tools/proto_blk/run_blk.py): level-1 32 -> 32 SubM conv over a BLOCK-LOCAL row order
// (DESIGN.md R3.6 "what comes next").  Rows are sorted by 8x8x8 block and cut into units of <= 64 own rows; a WAVE owns a unit end to end:
//   stage  : own rows (contiguous) + halo rows (gathered once each, present by construction) -> per-wave LDS stage by LDS-DMA, 64 B per row,
//            16-B pieces swizzled by (staged position >> 2) & 3;
//   taps   : the unit's local rulebook (one byte per (row, tap): staged position or 255 = absent -> an all-zero stage row) gives every A
//            fragment's LDS address; all 27 weight matrices are resident in LDS; 16x16x32 MFMAs, fp32 accumulators in registers;
//   output : accumulators -> stage area -> 16-B stores of the unit's own rows.
// No workgroup barrier after the weights are staged, no minima, no windows, no gather slot for an absent tap.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr;

static __device__ __forceinline__ u32x4 lds_r128(unsigned a) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); return v; }
#define LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory")
#define KEEP(x) asm volatile("" : "+v"(x))
static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  uint32_t a = __float_as_uint(lo), b = __float_as_uint(hi);
  a += 0x7FFFu + ((a >> 16) & 1u); b += 0x7FFFu + ((b >> 16) & 1u);
  return (a >> 16) | (b & 0xFFFF0000u);
}
static __device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct BlkP {
  const uint16_t* x;        // [n][32] bf16, block-local row order
  const uint16_t* w;        // [27][32 cout][32 cin] bf16
  uint16_t* out;            // [n][32] bf16
  const int32_t* unit;      // [nunits][4] = row0, n_own, halo offset, n_halo (halo list padded to a multiple of 16 with -1)
  const int32_t* halo;      // new-row indices
  const uint8_t* lrb;       // [nunits][4 groups][16 rows][32]: byte k = staged position of tap k, 255 = absent
  int64_t n;
  int nunits;
  int dbg;
};

constexpr int STAGE_B = 256 * 64;          // 256 stage rows of 64 B; row 255 stays zero
constexpr int WS_B = 27 * 32 * 64;

template <int NG>
static __device__ __forceinline__ void taps(const unsigned ws_a, const unsigned st_a, const unsigned b_off, const int pc, const u32x4 (&rb)[4][2],
                                            f32x4 (&acc)[4][2]) {
  // software pipeline of depth 2 over the taps: the reads of tap k + 1 are issued before the MFMAs of tap k (2 + NG reads per tap)
  u32x4 A[2][NG], B[2][2];
  auto issue = [&](int k, int s) __attribute__((always_inline)) {
    B[s][0] = lds_r128(ws_a + (unsigned)(k * 2048) + b_off);
    B[s][1] = lds_r128(ws_a + (unsigned)(k * 2048 + 1024) + b_off);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const uint32_t wd = rb[g][k >> 4][(k >> 2) & 3];
      const unsigned idx = (wd >> ((k & 3) * 8)) & 255u;
      A[s][g] = lds_r128(st_a + idx * 64u + (unsigned)(((pc ^ ((idx >> 2) & 3)) * 16)));
    }
  };
  issue(0, 0);
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int s = k & 1;
    if (k + 1 < 27) { issue(k + 1, s ^ 1); LGKM(2 + NG); } else LGKM(0);
    KEEP(B[s][0]); KEEP(B[s][1]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      KEEP(A[s][g]);
      acc[g][0] = mfma16(A[s][g], B[s][0], acc[g][0]);
      acc[g][1] = mfma16(A[s][g], B[s][1], acc[g][1]);
    }
  }
}

template <int W>
__global__ void __launch_bounds__(W * 64) k_conv_blk(BlkP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ri = lane & 15, pc = lane >> 4;
  {
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);
    for (int v = tid; v < 27 * 128; v += W * 64) {
      const int s = v & 3, n = (v >> 2) & 31, k = v >> 7;
      *reinterpret_cast<u32x4*>(smem + (k * 32 + n) * 64 + ((s ^ ((n >> 2) & 3)) * 16)) = wsrc[v];
    }
  }
  char* stage = smem + WS_B + wv * STAGE_B;
  for (int e = lane; e < STAGE_B / 16; e += 64) *reinterpret_cast<u32x4*>(stage + e * 16) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem, ws_a = lds0, st_a = lds0 + (unsigned)(WS_B + wv * STAGE_B);
  const unsigned b_off = (unsigned)(ri * 64 + ((pc ^ ((ri >> 2) & 3)) * 16));
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, (int)(p.n * 64), 0x00020000);
  const int gw = (int)blockIdx.x * W + wv, nw = (int)gridDim.x * W;

  for (int u = gw; u < p.nunits; u += nw) {
    const int row0 = p.unit[4 * u], nown = p.unit[4 * u + 1], h0 = p.unit[4 * u + 2], nh = p.unit[4 * u + 3];
    // 1. local rulebook of the unit: lane (ri, pc) takes the 32-byte records of rows 16 g + ri (the four pc groups load the same bytes)
    u32x4 rb[4][2];
    const u32x4* lr = reinterpret_cast<const u32x4*>(p.lrb + (int64_t)u * 2048);
#pragma unroll
    for (int g = 0; g < 4; ++g) { rb[g][0] = lr[(g * 16 + ri) * 2]; rb[g][1] = lr[(g * 16 + ri) * 2 + 1]; }
    // 2. stage: own rows, then halo rows
    if (!(p.dbg & 1)) {
      for (int c = 0; c * 16 < nown; ++c) {
        const int pos = c * 16 + (lane >> 2);
        const unsigned off = (unsigned)(row0 + pos) * 64u + (unsigned)((((lane & 3) ^ ((pos >> 2) & 3))) * 16);
        if (pos < nown) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + c * 1024), 16, (int)off, 0, 0, 0);   // (inactive lanes write nothing: the halo starts right behind)
      }
      for (int c = 0; c * 16 < nh; ++c) {
        const int hi = p.halo[h0 + c * 16 + (lane >> 2)];
        const int pos = nown + c * 16 + (lane >> 2);
        const unsigned off = hi >= 0 ? (unsigned)hi * 64u + (unsigned)((((lane & 3) ^ ((pos >> 2) & 3))) * 16) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + nown * 64 + c * 1024), 16, (int)off, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < 4; ++g) { KEEP(rb[g][0]); KEEP(rb[g][1]); }
    // 3. taps
    f32x4 acc[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) { acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int ng = (nown + 15) >> 4;
    if (!(p.dbg & 2)) {
      if (ng == 1) taps<1>(ws_a, st_a, b_off, pc, rb, acc);
      else if (ng == 2) taps<2>(ws_a, st_a, b_off, pc, rb, acc);
      else if (ng == 3) taps<3>(ws_a, st_a, b_off, pc, rb, acc);
      else taps<4>(ws_a, st_a, b_off, pc, rb, acc);
    }
    // 4. output through the stage (the halo part is dead now; rows 0 .. 63 x 36 floats = 9216 B), then the stage's used rows back to zero
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* ew = reinterpret_cast<float*>(stage);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ew[(g * 16 + 4 * pc + r) * 36 + ri] = acc[g][0][r];
        ew[(g * 16 + 4 * pc + r) * 36 + 16 + ri] = acc[g][1][r];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32x4 o[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = (lane >> 2) + 16 * it, cvv = lane & 3;
      const unsigned ea = st_a + (unsigned)((rr * 36 + cvv * 8) * 4);
      u32x4 u0 = lds_r128(ea), u1 = lds_r128(ea + 16);
      LGKM(0);
      KEEP(u0); KEEP(u1);
      o[it] = u32x4{pack2(__uint_as_float(u0[0]), __uint_as_float(u0[1])), pack2(__uint_as_float(u0[2]), __uint_as_float(u0[3])),
                    pack2(__uint_as_float(u1[0]), __uint_as_float(u1[1])), pack2(__uint_as_float(u1[2]), __uint_as_float(u1[3]))};
    }
    // rows 0 .. 255 of the stage minus the zero row: zero what the next unit may read as "absent" -- everything is rewritten by the next DMA
    // except the tail beyond its own + halo rows, which only matters for position 255 (never written).  The fp32 scratch overwrote rows 0..143:
    // the next unit's DMA overwrites its first n_own + n_halo rows and never reads beyond them, so nothing needs clearing.
    if (!(p.dbg & 4)) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rr = (lane >> 2) + 16 * it, cvv = lane & 3;
        if (rr < nown) *reinterpret_cast<u32x4*>(p.out + (int64_t)(row0 + rr) * 32 + cvv * 8) = o[it];
      }
    }
  }
}


// ---- variant R: the weights live in REGISTERS (27 taps x 2 cout halves x 16 B per lane = 216 registers; with one wave per SIMD the unified
// VGPR/AGPR file has 512 per lane and MFMA sources may sit in either half), so the LDS holds only stages: two per wave, and the rulebook +
// staging DMA of unit t + 1 are issued before the taps of unit t.
template <int NG>
static __device__ __forceinline__ void taps_r(const unsigned st_a, const int pc, const u32x4 (&rb)[4][2], const u32x4 (&Bw)[27][2], f32x4 (&acc)[4][2]) {
  constexpr int D = 4;                                     // taps in flight: with one wave per SIMD the LDS latency is covered by the wave's own MFMAs only
  u32x4 A[D][NG];
  auto issue = [&](int k, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const uint32_t wd = rb[g][k >> 4][(k >> 2) & 3];
      const unsigned idx = (wd >> ((k & 3) * 8)) & 255u;
      A[s][g] = lds_r128(st_a + idx * 64u + (unsigned)(((pc ^ ((idx >> 2) & 3)) * 16)));
    }
  };
#pragma unroll
  for (int k = 0; k < D - 1; ++k) issue(k, k);
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int s = k % D;
    if (k + D - 1 < 27) { issue(k + D - 1, (k + D - 1) % D); LGKM((D - 1) * NG); }
    else if (26 - k == 2) LGKM(2 * NG); else if (26 - k == 1) LGKM(NG); else LGKM(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      KEEP(A[s][g]);
      acc[g][0] = mfma16(A[s][g], Bw[k][0], acc[g][0]);
      acc[g][1] = mfma16(A[s][g], Bw[k][1], acc[g][1]);
    }
  }
}

template <int W>
__global__ void __launch_bounds__(W * 64) k_conv_blk_r(BlkP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ri = lane & 15, pc = lane >> 4;
  u32x4 Bw[27][2];
  {
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);            // [27][32 cout][4 pieces]
#pragma unroll
    for (int k = 0; k < 27; ++k) { Bw[k][0] = wsrc[(k * 32 + ri) * 4 + pc]; Bw[k][1] = wsrc[(k * 32 + 16 + ri) * 4 + pc]; }
  }
  char* stage0 = smem + wv * (2 * STAGE_B);
  for (int e = lane; e < 2 * STAGE_B / 16; e += 64) *reinterpret_cast<u32x4*>(stage0 + e * 16) = u32x4{0u, 0u, 0u, 0u};
  const unsigned st0 = (unsigned)(uintptr_t)(lds_ptr)stage0;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, (int)(p.n * 64), 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(p.n * 64), 0x00020000);
  const int gw = (int)blockIdx.x * W + wv, nw = (int)gridDim.x * W;

  auto stage_unit = [&](int u, char* stage) __attribute__((always_inline)) {
    const int row0 = p.unit[4 * u], nown = p.unit[4 * u + 1], h0 = p.unit[4 * u + 2], nh = p.unit[4 * u + 3];
    for (int c = 0; c * 16 < nown; ++c) {
      const int pos = c * 16 + (lane >> 2);
      const unsigned off = (unsigned)(row0 + pos) * 64u + (unsigned)((((lane & 3) ^ ((pos >> 2) & 3))) * 16);
      if (pos < nown) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + c * 1024), 16, (int)off, 0, 0, 0);
    }
    for (int c = 0; c * 16 < nh; ++c) {
      const int hi = p.halo[h0 + c * 16 + (lane >> 2)];
      const int pos = nown + c * 16 + (lane >> 2);
      const unsigned off = hi >= 0 ? (unsigned)hi * 64u + (unsigned)((((lane & 3) ^ ((pos >> 2) & 3))) * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + nown * 64 + c * 1024), 16, (int)off, 0, 0, 0);
    }
  };
  // (inline assembly: a compiler-visible load behind a run-time number of DMA instructions is waited for with vmcnt(0), which would also wait
  // for the previous unit's stores)
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.lrb), 0, (int)((int64_t)p.nunits * 2048), 0x00020000);
  auto load_rb = [&](int u, u32x4 (&rb)[4][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const unsigned off = (unsigned)u * 2048u + (unsigned)((g * 16 + ri) * 32);
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rb[g][0]) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(rb[g][1]) : "v"(off), "s"(rl));
    }
  };

  u32x4 rbc[4][2], rbn[4][2];
  int buf = 0;
  if (gw < p.nunits) { load_rb(gw, rbc); if (!(p.dbg & 1)) stage_unit(gw, stage0); }
  for (int u = gw; u < p.nunits; u += nw) {
    const int row0 = p.unit[4 * u], nown = p.unit[4 * u + 1];
    char* stage = stage0 + buf * STAGE_B;
    const unsigned st_a = st0 + (unsigned)(buf * STAGE_B);
    // this unit's stage and rulebook have landed (issued one unit ago); the only younger operations are the four stores of the previous unit
    if (u == gw) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (u != gw) {
#pragma unroll
      for (int g = 0; g < 4; ++g) { KEEP(rbn[g][0]); KEEP(rbn[g][1]); rbc[g][0] = rbn[g][0]; rbc[g][1] = rbn[g][1]; }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) { KEEP(rbc[g][0]); KEEP(rbc[g][1]); }
    }
    const int un = u + nw;
    if (un < p.nunits) { load_rb(un, rbn); if (!(p.dbg & 1)) stage_unit(un, stage0 + (buf ^ 1) * STAGE_B); }
    f32x4 acc[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) { acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int ng = (nown + 15) >> 4;
    if (!(p.dbg & 2)) {
      if (ng == 1) taps_r<1>(st_a, pc, rbc, Bw, acc);
      else if (ng == 2) taps_r<2>(st_a, pc, rbc, Bw, acc);
      else if (ng == 3) taps_r<3>(st_a, pc, rbc, Bw, acc);
      else taps_r<4>(st_a, pc, rbc, Bw, acc);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* ew = reinterpret_cast<float*>(stage);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ew[(g * 16 + 4 * pc + r) * 36 + ri] = acc[g][0][r];
        ew[(g * 16 + 4 * pc + r) * 36 + 16 + ri] = acc[g][1][r];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32x4 o[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = (lane >> 2) + 16 * it, cvv = lane & 3;
      const unsigned ea = st_a + (unsigned)((rr * 36 + cvv * 8) * 4);
      u32x4 u0 = lds_r128(ea), u1 = lds_r128(ea + 16);
      LGKM(0);
      KEEP(u0); KEEP(u1);
      o[it] = u32x4{pack2(__uint_as_float(u0[0]), __uint_as_float(u0[1])), pack2(__uint_as_float(u0[2]), __uint_as_float(u0[3])),
                    pack2(__uint_as_float(u1[0]), __uint_as_float(u1[1])), pack2(__uint_as_float(u1[2]), __uint_as_float(u1[3]))};
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {                                   // always four store instructions (a row past the unit: offset out of range, dropped)
      const int rr = (lane >> 2) + 16 * it, cvv = lane & 3;
      const unsigned off = (rr < nown && !(p.dbg & 4)) ? (unsigned)(row0 + rr) * 64u + (unsigned)(cvv * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o[it]), ro, (int)off, 0, 0);
    }
    buf ^= 1;
  }
}


// ---- variant S: variant R plus (1) 32x32x16 MFMAs on units of exactly 64 consecutive rows of the block-sorted order (two 32-row tiles; units may
// span blocks), (2) a 16-bit local rulebook that stores the swizzled stage offset (address = stage + (entry ^ 16 * piece)), (3) the halo indices
// of unit t + 2 requested while unit t is computed, so that the staging DMA of unit t + 1 never waits for its own indices.
struct BlkS {
  const uint16_t* x; const uint16_t* w; uint16_t* out;
  const int32_t* unit;      // [nunits][4] = row0, n_own, halo offset (in entries; lists padded to 160 entries), n_halo
  const int32_t* halo;
  const uint16_t* lrb;      // [nunits][64 rows][32]: entry k = stage byte offset of tap k (position * 64 + ((position >> 2) & 3) * 16), absent = 255 * 64
  int64_t n; int nunits; int dbg;
};
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HCH = 10;                     // halo chunks of 16 rows per unit (<= 160 halo rows)

template <int W>
__global__ void __launch_bounds__(W * 64) k_conv_blk_s(BlkS p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, fh = lane >> 5;
  u32x4 Bw[27][2];
  {
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);            // [27][32 cout][4 pieces]; k-step s, lane half fh: piece 2 s + fh
#pragma unroll
    for (int k = 0; k < 27; ++k) { Bw[k][0] = wsrc[(k * 32 + fi) * 4 + fh]; Bw[k][1] = wsrc[(k * 32 + fi) * 4 + 2 + fh]; }
  }
  char* stage0 = smem + wv * (2 * STAGE_B);
  for (int e = lane; e < 2 * STAGE_B / 16; e += 64) *reinterpret_cast<u32x4*>(stage0 + e * 16) = u32x4{0u, 0u, 0u, 0u};
  const unsigned st0 = (unsigned)(uintptr_t)(lds_ptr)stage0;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, (int)(p.n * 64), 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(p.n * 64), 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.lrb), 0, (int)((int64_t)p.nunits * 4096), 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.halo), 0, (int)((int64_t)p.nunits * HCH * 16 * 4), 0x00020000);
  const int gw = (int)blockIdx.x * W + wv, nw = (int)gridDim.x * W;
  const unsigned pc16[2] = {(unsigned)(fh * 16), (unsigned)(32 + fh * 16)};          // piece 2 s + fh

  // all loads of the steady state are inline assembly with fixed counts (HCH index loads, 8 rulebook loads per unit)
  auto load_hidx = [&](int u, int (&h)[HCH]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < HCH; ++c) {
      const unsigned off = u < p.nunits ? ((unsigned)u * (HCH * 16) + (unsigned)(c * 16 + (lane >> 2))) * 4u : 0xFFFFFFFFu;
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(h[c]) : "v"(off), "s"(rh));
    }
  };
  auto load_rb = [&](int u, u32x4 (&rb)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const unsigned off = u < p.nunits ? (unsigned)u * 4096u + (unsigned)((t * 32 + fi) * 64) : 0xFFFFFFFFu;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rb[t][0]) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(rb[t][1]) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:32" : "=v"(rb[t][2]) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:48" : "=v"(rb[t][3]) : "v"(off), "s"(rl));
    }
  };
  // staging DMA of a unit: 4 own chunks + HCH halo chunks, ALWAYS 4 + HCH instructions (rows past the unit / absent halo entries: out of range -> zeros)
  auto stage_unit = [&](int u, const int (&h)[HCH], char* stage) __attribute__((always_inline)) {
    const bool ok = u < p.nunits;
    const int row0 = ok ? p.unit[4 * u] : 0, nown = ok ? p.unit[4 * u + 1] : 0, nh = ok ? p.unit[4 * u + 3] : 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int pos = c * 16 + (lane >> 2);
      const unsigned off = pos < nown ? (unsigned)(row0 + pos) * 64u + (unsigned)((((lane & 3) ^ ((pos >> 2) & 3))) * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + c * 1024), 16, (int)off, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < HCH; ++c) {
      const int j = c * 16 + (lane >> 2), pos = 64 + j;                             // halo rows are staged from position 64 on
      const unsigned off = j < nh ? (unsigned)h[c] * 64u + (unsigned)((((lane & 3) ^ ((pos >> 2) & 3))) * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + 4096 + c * 1024), 16, (int)off, 0, 0, 0);
    }
  };

  int hc[HCH], hn[HCH];
  u32x4 rbc[2][4], rbn[2][4];
  load_hidx(gw, hc); load_hidx(gw + nw, hn); load_rb(gw, rbc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int c = 0; c < HCH; ++c) { KEEP(hc[c]); KEEP(hn[c]); }
  if (!(p.dbg & 1)) stage_unit(gw, hc, stage0);
  int buf = 0;
  bool firstu = true;
  for (int u = gw; u < p.nunits; u += nw) {
    const int row0 = p.unit[4 * u], nown = p.unit[4 * u + 1];
    char* stage = stage0 + buf * STAGE_B;
    const unsigned st_a = st0 + (unsigned)(buf * STAGE_B);
    // everything but the previous unit's four stores has landed: this unit's stage and rulebook, the next unit's halo indices
    if (firstu) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (!firstu) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) { KEEP(rbn[t][q]); rbc[t][q] = rbn[t][q]; }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) KEEP(rbc[t][q]);
    }
    firstu = false;
    // requests for the units behind: indices of unit t + 2, rulebook and stage of unit t + 1
    int hnn[HCH];
    load_hidx(u + 2 * nw, hnn);
    load_rb(u + nw, rbn);
    if (!(p.dbg & 1)) stage_unit(u + nw, hn, stage0 + (buf ^ 1) * STAGE_B);

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    if (!(p.dbg & 2)) {
      constexpr int D = 3;
      u32x4 A[D][2][2];
      auto issue = [&](int k, int s_) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint32_t wd = rbc[t][k >> 3][(k >> 1) & 3];
          const unsigned val = (k & 1) ? (wd >> 16) : (wd & 0xFFFFu);
          const unsigned a0 = st_a + val;
          A[s_][t][0] = lds_r128(a0 ^ pc16[0]);
          A[s_][t][1] = lds_r128(a0 ^ pc16[1]);
        }
      };
#pragma unroll
      for (int k = 0; k < D - 1; ++k) issue(k, k);
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const int s_ = k % D;
        if (k + D - 1 < 27) { issue(k + D - 1, (k + D - 1) % D); LGKM((D - 1) * 4); }
        else if (26 - k == 1) LGKM(4); else LGKM(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          KEEP(A[s_][t][0]); KEEP(A[s_][t][1]);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[s_][t][0]), __builtin_bit_cast(bf16x8, Bw[k][0]), acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[s_][t][1]), __builtin_bit_cast(bf16x8, Bw[k][1]), acc[t], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* ew = reinterpret_cast<float*>(stage);                                    // 64 rows x 36 floats over the dead stage
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) ew[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 36 + fi] = acc[t][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32x4 o[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = (lane >> 2) + 16 * it, cvv = lane & 3;
      const unsigned ea = st_a + (unsigned)((rr * 36 + cvv * 8) * 4);
      u32x4 u0 = lds_r128(ea), u1 = lds_r128(ea + 16);
      LGKM(0);
      KEEP(u0); KEEP(u1);
      o[it] = u32x4{pack2(__uint_as_float(u0[0]), __uint_as_float(u0[1])), pack2(__uint_as_float(u0[2]), __uint_as_float(u0[3])),
                    pack2(__uint_as_float(u1[0]), __uint_as_float(u1[1])), pack2(__uint_as_float(u1[2]), __uint_as_float(u1[3]))};
    }
    // the zero row (position 255) lies beyond the scratch; the scratch itself is overwritten by the next unit's DMA into this buffer
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = (lane >> 2) + 16 * it, cvv = lane & 3;
      const unsigned off = (rr < nown && !(p.dbg & 4)) ? (unsigned)(row0 + rr) * 64u + (unsigned)(cvv * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o[it]), ro, (int)off, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < HCH; ++c) { hn[c] = hnn[c]; }
    buf ^= 1;
  }
}

extern "C" int conv_blk_s(const void* x, const void* w, void* out, const void* unit, const void* halo, const void* lrb, int64_t n, int nunits, int waves,
                          int dbg, void* stream) {
  BlkS p{(const uint16_t*)x, (const uint16_t*)w, (uint16_t*)out, (const int32_t*)unit, (const int32_t*)halo, (const uint16_t*)lrb, n, nunits, dbg};
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)waves * 2 * STAGE_B;
  const int per_cu = 4 / waves;
#define GOS(WW)                                                                                                                  \
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_blk_s<WW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; \
  k_conv_blk_s<WW><<<256 * per_cu, WW * 64, lds, s>>>(p);
  if (waves == 4) { GOS(4) } else if (waves == 2) { GOS(2) } else if (waves == 1) { GOS(1) } else return -2;
#undef GOS
  return hipGetLastError() == hipSuccess ? 0 : -3;
}


// ---- variant T: level 2, 64 -> 64 channels (128-B rows).  A workgroup of four waves (one per SIMD) shares the stage of a unit of <= 64 own rows +
// <= 192 halo rows; wave w computes output channels 16 w .. 16 w + 15 with 16x16x32 MFMAs and keeps ITS slice of all 27 weight matrices in
// registers (27 taps x 2 k-steps x 16 B = 216 registers per lane).  One barrier per unit (the stage is filled by all four waves), two stages.
struct BlkT {
  const uint16_t* x; const uint16_t* w; uint16_t* out;
  const int32_t* unit;      // [nunits][4] = row0, n_own, -, n_halo
  const int32_t* halo;      // [nunits][192]
  const uint16_t* lrb;      // [nunits][64 rows][32]: entry k = stage byte offset of tap k (position * 128 + ((position >> 1) & 7) * 16), absent = 511 * 128
  int64_t n; int nunits; int dbg;
};
constexpr int T_STAGE = 512 * 128;            // 512 stage rows of 128 B would be 64 KB; we use 320 rows (40 KB) per stage, zero row at position 319
constexpr int T_ROWS = 320, T_STAGE_B = T_ROWS * 128, T_HALO = 192, T_ZERO = 319;

__global__ void __launch_bounds__(256) k_conv_blk_t(BlkT p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ri = lane & 15, kq = lane >> 4;
  u32x4 Bw[27][2];
  {
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);            // [27][64 cout][8 pieces of 8 cin]
#pragma unroll
    for (int k = 0; k < 27; ++k) { Bw[k][0] = wsrc[(k * 64 + 16 * wv + ri) * 8 + kq]; Bw[k][1] = wsrc[(k * 64 + 16 * wv + ri) * 8 + 4 + kq]; }
  }
  for (int e = tid; e < 2 * T_STAGE_B / 16; e += 256) *reinterpret_cast<u32x4*>(smem + e * 16) = u32x4{0u, 0u, 0u, 0u};
  float* escr = reinterpret_cast<float*>(smem + 2 * T_STAGE_B) + wv * (64 * 20);            // per-wave epilogue scratch [64 rows][16 + 4]
  __syncthreads();
  const unsigned st0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned es_a = st0 + (unsigned)(2 * T_STAGE_B + wv * (64 * 20 * 4));
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, (int)(p.n * 128), 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(p.n * 128), 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.lrb), 0, (int)((int64_t)p.nunits * 4096), 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.halo), 0, (int)((int64_t)p.nunits * T_HALO * 4), 0x00020000);
  const int gb = (int)blockIdx.x, nb = (int)gridDim.x;
  const unsigned pq16[2] = {(unsigned)(kq * 16), (unsigned)(64 + kq * 16)};            // k-step s, lane quarter kq: piece 4 s + kq

  // wave w stages own chunks 2 w, 2 w + 1 (8 rows of 128 B each) and halo chunks 6 w .. 6 w + 5: always 8 DMA instructions, 6 index loads
  auto load_hidx = [&](int u, int (&h)[6]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const unsigned off = u < p.nunits ? ((unsigned)u * T_HALO + (unsigned)((6 * wv + c) * 8 + (lane >> 3))) * 4u : 0xFFFFFFFFu;
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(h[c]) : "v"(off), "s"(rh));
    }
  };
  auto load_rb = [&](int u, u32x4 (&rb)[4][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const unsigned off = u < p.nunits ? (unsigned)u * 4096u + (unsigned)((g * 16 + ri) * 64) : 0xFFFFFFFFu;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rb[g][0]) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(rb[g][1]) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:32" : "=v"(rb[g][2]) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:48" : "=v"(rb[g][3]) : "v"(off), "s"(rl));
    }
  };
  auto stage_unit = [&](int u, const int (&h)[6], char* stage) __attribute__((always_inline)) {
    const bool ok = u < p.nunits;
    const int row0 = ok ? p.unit[4 * u] : 0, nown = ok ? p.unit[4 * u + 1] : 0, nh = ok ? p.unit[4 * u + 3] : 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int pos = (2 * wv + c) * 8 + (lane >> 3);
      const unsigned off = pos < nown ? (unsigned)(row0 + pos) * 128u + (unsigned)((((lane & 7) ^ ((pos >> 1) & 7))) * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + (2 * wv + c) * 1024), 16, (int)off, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int j = (6 * wv + c) * 8 + (lane >> 3), pos = 64 + j;
      const unsigned off = j < nh ? (unsigned)h[c] * 128u + (unsigned)((((lane & 7) ^ ((pos >> 1) & 7))) * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + 8192 + (6 * wv + c) * 1024), 16, (int)off, 0, 0, 0);
    }
  };

  int hc[6], hn[6];
  u32x4 rbc[4][4], rbn[4][4];
  load_hidx(gb, hc); load_hidx(gb + nb, hn); load_rb(gb, rbc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int c = 0; c < 6; ++c) { KEEP(hc[c]); KEEP(hn[c]); }
  if (!(p.dbg & 1)) stage_unit(gb, hc, smem);
  int buf = 0;
  bool firstu = true;
  for (int u = gb; u < p.nunits; u += nb) {
    const int row0 = p.unit[4 * u], nown = p.unit[4 * u + 1];
    const unsigned st_a = st0 + (unsigned)(buf * T_STAGE_B);
    // this wave's share of the unit's stage, its rulebook and the next unit's indices have landed (the two stores of the previous unit may be
    // outstanding); the barrier makes the other waves' shares visible and says that everybody is done with the other stage
    if (firstu) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (!firstu) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) { KEEP(rbn[g][q]); rbc[g][q] = rbn[g][q]; }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) KEEP(rbc[g][q]);
    }
    firstu = false;
    int hnn[6];
    load_hidx(u + 2 * nb, hnn);
    load_rb(u + nb, rbn);
    if (!(p.dbg & 1)) stage_unit(u + nb, hn, smem + (buf ^ 1) * T_STAGE_B);

    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(p.dbg & 2)) {
      constexpr int D = 2;                               // (lgkmcnt counts to 15: one tap of 8 reads ahead)
      u32x4 A[D][4][2];
      auto issue = [&](int k, int s_) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t wd = rbc[g][k >> 3][(k >> 1) & 3];
          const unsigned val = (k & 1) ? (wd >> 16) : (wd & 0xFFFFu);
          const unsigned a0 = st_a + val;
          A[s_][g][0] = lds_r128(a0 ^ pq16[0]);
          A[s_][g][1] = lds_r128(a0 ^ pq16[1]);
        }
      };
#pragma unroll
      for (int k = 0; k < D - 1; ++k) issue(k, k);
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const int s_ = k % D;
        if (k + D - 1 < 27) { issue(k + D - 1, (k + D - 1) % D); LGKM((D - 1) * 8); }
        else LGKM(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          KEEP(A[s_][g][0]); KEEP(A[s_][g][1]);
          acc[g] = mfma16(A[s_][g][0], Bw[k][0], acc[g]);
          acc[g] = mfma16(A[s_][g][1], Bw[k][1], acc[g]);
        }
      }
    }
    // epilogue: the wave's [64 rows][16 channels] through its own scratch -> 16-B stores (two per lane)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) escr[(g * 16 + 4 * kq + r) * 20 + ri] = acc[g][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int rr = (lane >> 1) + 32 * it, cv = lane & 1;
      const unsigned ea = es_a + (unsigned)((rr * 20 + cv * 8) * 4);
      u32x4 u0 = lds_r128(ea), u1 = lds_r128(ea + 16);
      LGKM(0);
      KEEP(u0); KEEP(u1);
      const u32x4 o = u32x4{pack2(__uint_as_float(u0[0]), __uint_as_float(u0[1])), pack2(__uint_as_float(u0[2]), __uint_as_float(u0[3])),
                            pack2(__uint_as_float(u1[0]), __uint_as_float(u1[1])), pack2(__uint_as_float(u1[2]), __uint_as_float(u1[3]))};
      const unsigned off = (rr < nown && !(p.dbg & 4)) ? (unsigned)(row0 + rr) * 128u + (unsigned)(wv * 32 + cv * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), ro, (int)off, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) hn[c] = hnn[c];
    buf ^= 1;
  }
}

extern "C" int conv_blk_t(const void* x, const void* w, void* out, const void* unit, const void* halo, const void* lrb, int64_t n, int nunits, int dbg,
                          void* stream) {
  BlkT p{(const uint16_t*)x, (const uint16_t*)w, (uint16_t*)out, (const int32_t*)unit, (const int32_t*)halo, (const uint16_t*)lrb, n, nunits, dbg};
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)2 * T_STAGE_B + 4 * 64 * 20 * 4;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_blk_t), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
  k_conv_blk_t<<<256, 256, lds, s>>>(p);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}


// ---- variant U: level 2, 64 -> 64, "stream-q with a stage": a workgroup of two waves (two workgroups per CU) owns a unit of <= 64 rows (wave w: rows
// 32 w .. 32 w + 31), the unit's own + halo rows are staged once in LDS (two stages), the weights are streamed tap by tap through a
// double-buffered 8-KB LDS tile by LDS-DMA (one barrier per tap, as in tl_conv_streamq.hip), 32x32x16 MFMAs: per tap and wave 4 A fragments
// from the stage + 8 B fragments from the weight tile feed 8 MFMAs.  The staging DMAs of the next unit are interleaved with the taps (one per
// tap after that tap's weight DMAs) so that the per-tap counted wait never covers them.
struct BlkU {
  const uint16_t* x; const uint16_t* w; uint16_t* out;
  const int32_t* unit;      // [nunits][4] = row0, n_own, -, n_halo (<= 184)
  const int32_t* halo;      // [nunits][192]
  const uint16_t* lrb;      // [nunits][64 rows][32]: stage byte offset (position * 128 + ((position >> 1) & 7) * 16), absent = 255 * 128
  int64_t n; int nunits; int dbg;
};
constexpr int U_STAGE_B = 256 * 128, U_W_B = 64 * 128;

__global__ void __launch_bounds__(128) k_conv_blk_u(BlkU p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];       // [2 stages][256 rows][128 B] | [2][64 cout][128 B] weights
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, fh = lane >> 5;
  for (int e = tid; e < 2 * U_STAGE_B / 16; e += 128) *reinterpret_cast<u32x4*>(smem + e * 16) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  const unsigned st0 = (unsigned)(uintptr_t)(lds_ptr)smem, wb0 = st0 + 2 * U_STAGE_B;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, (int)(p.n * 128), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, 27 * 64 * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(p.n * 128), 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.lrb), 0, (int)((int64_t)p.nunits * 4096), 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.halo), 0, (int)((int64_t)p.nunits * 192 * 4), 0x00020000);
  const int gb = (int)blockIdx.x, nb = (int)gridDim.x;
  // weight DMA: tap k = 8 KB = 8 chunks of 8 cout rows; wave w takes chunks 4 w .. 4 w + 3; lane L: row L >> 3, position L & 7 holds piece (L & 7) ^ (row & 7)
  const unsigned wdma = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16));
  auto dma_w = [&](int k, int wbuf) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ch = 4 * wv + c;
      const unsigned off = k < 27 ? (unsigned)(k * U_W_B + ch * 1024) + wdma : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + 2 * U_STAGE_B + wbuf * U_W_B + ch * 1024), 16, (int)off, 0, 0, 0);
    }
  };
  // B fragment of (cout block nbk, k-step s): lane (n = fi, half fh): row 32 nbk + fi, piece 2 s + fh at position (2 s + fh) ^ (fi & 7)
  unsigned boff[4];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) boff[s_] = (unsigned)(fi * 128 + (((2 * s_ + fh) ^ (fi & 7)) * 16));
  // staging: 32 chunks of 8 rows per unit (8 own + 24 halo); wave w issues chunks w, w + 2, ... (16 per wave), chunk j of the wave at tap j
  auto load_hidx = [&](int u, int (&h)[12]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      const unsigned off = u < p.nunits ? ((unsigned)u * 192u + (unsigned)((2 * c + wv) * 8 + (lane >> 3))) * 4u : 0xFFFFFFFFu;
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(h[c]) : "v"(off), "s"(rh));
    }
  };
  auto load_rb = [&](int u, u32x4 (&rb)[4]) __attribute__((always_inline)) {
    const unsigned off = u < p.nunits ? (unsigned)u * 4096u + (unsigned)((wv * 32 + fi) * 64) : 0xFFFFFFFFu;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rb[0]) : "v"(off), "s"(rl));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(rb[1]) : "v"(off), "s"(rl));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:32" : "=v"(rb[2]) : "v"(off), "s"(rl));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:48" : "=v"(rb[3]) : "v"(off), "s"(rl));
  };
  // one staging chunk (j = 0 .. 15 of this wave) of unit u into `stage`
  auto stage_chunk = [&](int u, int j, const int (&h)[12], char* stage) __attribute__((always_inline)) {
    const bool ok = u < p.nunits;
    const int row0 = ok ? p.unit[4 * u] : 0, nown = ok ? p.unit[4 * u + 1] : 0, nh = ok ? p.unit[4 * u + 3] : 0;
    const int ch = 2 * j + wv;                                                        // chunk 0 .. 31 of the unit
    const int pos = ch * 8 + (lane >> 3);
    unsigned off;
    if (j < 4) off = pos < nown ? (unsigned)(row0 + pos) * 128u + (unsigned)((((lane & 7) ^ ((pos >> 1) & 7))) * 16) : 0xFFFFFFFFu;
    else {
      const int hj = pos - 64;
      off = hj < nh ? (unsigned)h[j - 4] * 128u + (unsigned)((((lane & 7) ^ ((pos >> 1) & 7))) * 16) : 0xFFFFFFFFu;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + ch * 1024), 16, (int)off, 0, 0, 0);
  };

  int hc[12], hn[12];
  u32x4 rbc[4], rbn[4];
  load_hidx(gb, hc); load_hidx(gb + nb, hn); load_rb(gb, rbc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int c = 0; c < 12; ++c) { KEEP(hc[c]); KEEP(hn[c]); }
  if (!(p.dbg & 1)) {
#pragma unroll
    for (int j = 0; j < 16; ++j) stage_chunk(gb, j, hc, smem);
  }
  dma_w(0, 0);
  int buf = 0;
  bool firstu = true;
  for (int u = gb; u < p.nunits; u += nb) {
    const int row0 = p.unit[4 * u], nown = p.unit[4 * u + 1];
    const unsigned st_a = st0 + (unsigned)(buf * U_STAGE_B);
    char* nstage = smem + (buf ^ 1) * U_STAGE_B;
    // entry: outstanding = [this unit's stage chunks (issued during the previous unit's taps), rulebook / indices, weights of tap 0, the
    // previous unit's stores]: everything but the stores (4 per lane) has landed after this wait
    if (firstu) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (!firstu) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { KEEP(rbn[q]); rbc[q] = rbn[q]; }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) KEEP(rbc[q]);
    }
    firstu = false;
    int hnn[12];
    load_hidx(u + 2 * nb, hnn);
    load_rb(u + nb, rbn);
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      // weights of tap k have landed in this wave (waited for at the end of tap k - 1 / at entry); barrier: in the other wave too, and
      // both waves are done with weight buffer (k + 1) & 1 (tap k - 1)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      dma_w(k + 1, (k + 1) & 1);                                                     // (tap 27 = tap 0 of the next unit's weights: requested below instead)
      if (k < 16 && !(p.dbg & 1)) stage_chunk(u + nb, k, hn, nstage);               // one staging chunk of the next unit per tap, younger than the weights
      if (!(p.dbg & 2)) {
        const uint32_t wd = rbc[k >> 3][(k >> 1) & 3];
        const unsigned val = (k & 1) ? (wd >> 16) : (wd & 0xFFFFu);
        const unsigned a0 = st_a + val;
        u32x4 A[4], B[2][4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) A[s_] = lds_r128(a0 ^ (unsigned)((2 * s_ + fh) * 16));
        const unsigned wbb = wb0 + (unsigned)((k & 1) * U_W_B);
#pragma unroll
        for (int nbk = 0; nbk < 2; ++nbk)
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_) B[nbk][s_] = lds_r128(wbb + (unsigned)(nbk * 32 * 128) + boff[s_]);
        LGKM(0);
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) { KEEP(A[s_]); KEEP(B[0][s_]); KEEP(B[1][s_]); }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[s_]), __builtin_bit_cast(bf16x8, B[0][s_]), acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[s_]), __builtin_bit_cast(bf16x8, B[1][s_]), acc[1], 0, 0, 0);
        }
      }
      // weights of tap k + 1 must have landed before the next barrier; younger than them: at most this tap's staging chunk
      if (k < 16 && !(p.dbg & 1)) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (dma_w(27, 1) requested nothing: out-of-range offsets.)  Weights of the next unit's tap 0 into buffer 0: free after the barrier below.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    dma_w(0, 0);
    // epilogue through the dead stage: wave w rows 32 w .. 32 w + 31, 64 channels (pitch 68 floats)
    float* ew = reinterpret_cast<float*>(smem + buf * U_STAGE_B) + wv * (32 * 68);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) ew[((r & 3) + 8 * (r >> 2) + 4 * fh) * 68 + t * 32 + fi] = acc[t][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = (lane >> 3) + 8 * it, cv = lane & 7;
      const unsigned ea = st_a + (unsigned)((wv * 32 * 68 + rr * 68 + cv * 8) * 4);
      u32x4 u0 = lds_r128(ea), u1 = lds_r128(ea + 16);
      LGKM(0);
      KEEP(u0); KEEP(u1);
      const u32x4 o = u32x4{pack2(__uint_as_float(u0[0]), __uint_as_float(u0[1])), pack2(__uint_as_float(u0[2]), __uint_as_float(u0[3])),
                            pack2(__uint_as_float(u1[0]), __uint_as_float(u1[1])), pack2(__uint_as_float(u1[2]), __uint_as_float(u1[3]))};
      const int lr = wv * 32 + rr;
      const unsigned off = (lr < nown && !(p.dbg & 4)) ? (unsigned)(row0 + lr) * 128u + (unsigned)(cv * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), ro, (int)off, 0, 0);
    }
    // both waves must be out of the scratch before the next-but-one unit's staging overwrites it: that staging starts after the NEXT unit's
    // first barrier, which every wave reaches only after its epilogue
#pragma unroll
    for (int c = 0; c < 12; ++c) hn[c] = hnn[c];
    buf ^= 1;
  }
}

extern "C" int conv_blk_u(const void* x, const void* w, void* out, const void* unit, const void* halo, const void* lrb, int64_t n, int nunits, int dbg,
                          void* stream) {
  BlkU p{(const uint16_t*)x, (const uint16_t*)w, (uint16_t*)out, (const int32_t*)unit, (const int32_t*)halo, (const uint16_t*)lrb, n, nunits, dbg};
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)2 * U_STAGE_B + 2 * U_W_B;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_blk_u), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
  k_conv_blk_u<<<512, 128, lds, s>>>(p);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}


// ---- variant V: level 2, 64 -> 64: tl_conv_streamq.hip's structure (8 waves, weights streamed tap by tap through LDS, one barrier per tap, 32x32x16
// MFMAs, 4 A + 8 B fragments per 8 MFMAs) with the A fragments read from STAGED units instead of gathered: the workgroup owns four units at a time
// (wave pair q: unit 4 j + q, own single-buffered stage of 256 rows), a ring of four 8-KB weight tiles filled by LDS-DMA three taps ahead (one
// 1-KB chunk per wave and tap), the next four units' halo indices / rulebooks requested during the taps, their staging between the quads.
__global__ void __launch_bounds__(512) k_conv_blk_v(BlkU p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];       // [4 stages][256 rows][128 B] | [4][64 cout][128 B] weight ring
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, fh = lane >> 5, pq = wv >> 1, pw = wv & 1;           // pair, wave in pair
  for (int e = tid; e < 4 * U_STAGE_B / 16; e += 512) *reinterpret_cast<u32x4*>(smem + e * 16) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  const unsigned st0 = (unsigned)(uintptr_t)(lds_ptr)smem, wb0 = st0 + 4 * U_STAGE_B;
  char* stage = smem + pq * U_STAGE_B;
  const unsigned st_a = st0 + (unsigned)(pq * U_STAGE_B);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, (int)(p.n * 128), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, 27 * 64 * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(p.n * 128), 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.lrb), 0, (int)((int64_t)p.nunits * 4096), 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.halo), 0, (int)((int64_t)p.nunits * 192 * 4), 0x00020000);
  const int gb = (int)blockIdx.x, nb = (int)gridDim.x;
  const int nquads = (p.nunits + 3) >> 2;
  // weight DMA: wave w fills chunk w (cout rows 8 w .. 8 w + 7) of tap k's tile
  const unsigned wdma = (unsigned)(wv * 1024 + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16));
  auto dma_w = [&](int k) __attribute__((always_inline)) {
    const unsigned off = k < 27 ? (unsigned)(k * U_W_B) + wdma : 0xFFFFFFFFu;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + 4 * U_STAGE_B + (k & 3) * U_W_B + wv * 1024), 16, (int)off, 0, 0, 0);
  };
  unsigned boff[4];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) boff[s_] = (unsigned)(fi * 128 + (((2 * s_ + fh) ^ (fi & 7)) * 16));
  auto load_hidx = [&](int u, int (&h)[12]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      const unsigned off = u < p.nunits ? ((unsigned)u * 192u + (unsigned)((2 * c + pw) * 8 + (lane >> 3))) * 4u : 0xFFFFFFFFu;
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(h[c]) : "v"(off), "s"(rh));
    }
  };
  auto load_rb = [&](int u, u32x4 (&rb)[4]) __attribute__((always_inline)) {
    const unsigned off = u < p.nunits ? (unsigned)u * 4096u + (unsigned)((pw * 32 + fi) * 64) : 0xFFFFFFFFu;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rb[0]) : "v"(off), "s"(rl));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(rb[1]) : "v"(off), "s"(rl));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:32" : "=v"(rb[2]) : "v"(off), "s"(rl));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:48" : "=v"(rb[3]) : "v"(off), "s"(rl));
  };
  auto stage_unit = [&](int u, const int (&h)[12]) __attribute__((always_inline)) {       // 16 chunks of 8 rows per wave (the pair: 32)
    const bool ok = u < p.nunits;
    const int row0 = ok ? p.unit[4 * u] : 0, nown = ok ? p.unit[4 * u + 1] : 0, nh = ok ? p.unit[4 * u + 3] : 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ch = 2 * j + pw, pos = ch * 8 + (lane >> 3);
      unsigned off;
      if (j < 4) off = pos < nown ? (unsigned)(row0 + pos) * 128u + (unsigned)((((lane & 7) ^ ((pos >> 1) & 7))) * 16) : 0xFFFFFFFFu;
      else { const int hj = pos - 64; off = hj < nh ? (unsigned)h[j - 4] * 128u + (unsigned)((((lane & 7) ^ ((pos >> 1) & 7))) * 16) : 0xFFFFFFFFu; }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + ch * 1024), 16, (int)off, 0, 0, 0);
    }
  };

  int hc[12], hn[12];
  u32x4 rbc[4], rbn[4];
  load_hidx(4 * gb + pq, hc); load_rb(4 * gb + pq, rbc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int c = 0; c < 12; ++c) KEEP(hc[c]);
#pragma unroll
  for (int q = 0; q < 4; ++q) KEEP(rbc[q]);
  for (int j = gb; j < nquads; j += nb) {
    const int u = 4 * j + pq;
    const bool uok = u < p.nunits;
    const int row0 = uok ? p.unit[4 * u] : 0, nown = uok ? p.unit[4 * u + 1] : 0;
    // stage this quad (the stages are free: everybody left the previous quad's epilogue at the barrier below), weights of taps 0 .. 2
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (!(p.dbg & 1)) stage_unit(u, hc);
    dma_w(0); dma_w(1); dma_w(2);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                   // stage + weights(0) landed (stores of the previous quad are older than all of it)
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // weights(k) landed in every wave; ring slot (k + 3) & 3 (tap k - 1) is free
      dma_w(k + 3);
      if (k == 5) { load_hidx(4 * (j + nb) + pq, hn); load_rb(4 * (j + nb) + pq, rbn); }
      if (!(p.dbg & 2)) {
        const uint32_t wd = rbc[k >> 3][(k >> 1) & 3];
        const unsigned val = (k & 1) ? (wd >> 16) : (wd & 0xFFFFu);
        const unsigned a0 = st_a + val;
        u32x4 A[4], B[2][4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) A[s_] = lds_r128(a0 ^ (unsigned)((2 * s_ + fh) * 16));
        const unsigned wbb = wb0 + (unsigned)((k & 3) * U_W_B);
#pragma unroll
        for (int nbk = 0; nbk < 2; ++nbk)
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_) B[nbk][s_] = lds_r128(wbb + (unsigned)(nbk * 32 * 128) + boff[s_]);
        LGKM(0);
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) { KEEP(A[s_]); KEEP(B[0][s_]); KEEP(B[1][s_]); }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[s_]), __builtin_bit_cast(bf16x8, B[0][s_]), acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[s_]), __builtin_bit_cast(bf16x8, B[1][s_]), acc[1], 0, 0, 0);
        }
      }
      // before the next barrier: this wave's chunk of weights(k + 1); younger: weights(k + 2), (k + 3) and, for three taps, the 16 look-ahead loads
      if (k >= 5 && k <= 7) asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every pair is done with its stage
    float* ew = reinterpret_cast<float*>(stage) + pw * (32 * 68);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) ew[((r & 3) + 8 * (r >> 2) + 4 * fh) * 68 + t * 32 + fi] = acc[t][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = (lane >> 3) + 8 * it, cv = lane & 7;
      const unsigned ea = st_a + (unsigned)((pw * 32 * 68 + rr * 68 + cv * 8) * 4);
      u32x4 u0 = lds_r128(ea), u1 = lds_r128(ea + 16);
      LGKM(0);
      KEEP(u0); KEEP(u1);
      const u32x4 o = u32x4{pack2(__uint_as_float(u0[0]), __uint_as_float(u0[1])), pack2(__uint_as_float(u0[2]), __uint_as_float(u0[3])),
                            pack2(__uint_as_float(u1[0]), __uint_as_float(u1[1])), pack2(__uint_as_float(u1[2]), __uint_as_float(u1[3]))};
      const int lr = pw * 32 + rr;
      const unsigned off = (lr < nown && !(p.dbg & 4)) ? (unsigned)(row0 + lr) * 128u + (unsigned)(cv * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), ro, (int)off, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 12; ++c) { KEEP(hn[c]); hc[c] = hn[c]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) { KEEP(rbn[q]); rbc[q] = rbn[q]; }
  }
}

extern "C" int conv_blk_v(const void* x, const void* w, void* out, const void* unit, const void* halo, const void* lrb, int64_t n, int nunits, int dbg,
                          void* stream) {
  BlkU p{(const uint16_t*)x, (const uint16_t*)w, (uint16_t*)out, (const int32_t*)unit, (const int32_t*)halo, (const uint16_t*)lrb, n, nunits, dbg};
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)4 * U_STAGE_B + 4 * U_W_B;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_blk_v), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
  k_conv_blk_v<<<256, 512, lds, s>>>(p);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}


// ---- variant W: variant S's tap loop with TWO waves per SIMD: the weights go back to LDS (55 KB), every wave has ONE stage of 192 rows (halo <= 128), the
// next unit's staging is requested as soon as this unit's results have left the stage (before its stores), rulebook / indices one and two units ahead.
// While one wave of a SIMD waits for its stage, the other one computes.
constexpr int W_STAGE_B = 192 * 64, W_HCH = 8, W_ZERO = 191;

template <int W>
__global__ void __launch_bounds__(W * 64) k_conv_blk_w(BlkS p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, fh = lane >> 5;
  {
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);
    for (int v = tid; v < 27 * 128; v += W * 64) {
      const int s_ = v & 3, n = (v >> 2) & 31, k = v >> 7;
      *reinterpret_cast<u32x4*>(smem + (k * 32 + n) * 64 + ((s_ ^ ((n >> 2) & 3)) * 16)) = wsrc[v];
    }
  }
  char* stage = smem + WS_B + wv * W_STAGE_B;
  for (int e = lane; e < W_STAGE_B / 16; e += 64) *reinterpret_cast<u32x4*>(stage + e * 16) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem, st_a = lds0 + (unsigned)(WS_B + wv * W_STAGE_B);
  unsigned boff[2];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) boff[s_] = lds0 + (unsigned)(fi * 64 + (((2 * s_ + fh) ^ ((fi >> 2) & 3)) * 16));
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, (int)(p.n * 64), 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(p.n * 64), 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.lrb), 0, (int)((int64_t)p.nunits * 4096), 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.halo), 0, (int)((int64_t)p.nunits * W_HCH * 16 * 4), 0x00020000);
  const int gw = (int)blockIdx.x * W + wv, nw = (int)gridDim.x * W;
  const unsigned pc16[2] = {(unsigned)(fh * 16), (unsigned)(32 + fh * 16)};

  auto load_hidx = [&](int u, int (&h)[W_HCH]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < W_HCH; ++c) {
      const unsigned off = u < p.nunits ? ((unsigned)u * (W_HCH * 16) + (unsigned)(c * 16 + (lane >> 2))) * 4u : 0xFFFFFFFFu;
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(h[c]) : "v"(off), "s"(rh));
    }
  };
  auto load_rb = [&](int u, u32x4 (&rb)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const unsigned off = u < p.nunits ? (unsigned)u * 4096u + (unsigned)((t * 32 + fi) * 64) : 0xFFFFFFFFu;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rb[t][0]) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(rb[t][1]) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:32" : "=v"(rb[t][2]) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:48" : "=v"(rb[t][3]) : "v"(off), "s"(rl));
    }
  };
  // staging: 4 own chunks + as many halo chunks as the unit has (the counted wait below does not depend on their number)
  auto stage_unit = [&](int u, const int (&h)[W_HCH]) __attribute__((always_inline)) {
    if (u >= p.nunits) return;
    const int row0 = p.unit[4 * u], nown = p.unit[4 * u + 1], nh = p.unit[4 * u + 3];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int pos = c * 16 + (lane >> 2);
      const unsigned off = pos < nown ? (unsigned)(row0 + pos) * 64u + (unsigned)((((lane & 3) ^ ((pos >> 2) & 3))) * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + c * 1024), 16, (int)off, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < W_HCH; ++c) {
      if (c * 16 < nh) {
        const int j = c * 16 + (lane >> 2), pos = 64 + j;
        const unsigned off = j < nh ? (unsigned)h[c] * 64u + (unsigned)((((lane & 3) ^ ((pos >> 2) & 3))) * 16) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + 4096 + c * 1024), 16, (int)off, 0, 0, 0);
      }
    }
  };

  int hn[W_HCH];
  u32x4 rbc[2][4], rbn[2][4];
  {
    int h0[W_HCH];
    load_hidx(gw, h0); load_hidx(gw + nw, hn); load_rb(gw, rbc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < W_HCH; ++c) { KEEP(h0[c]); KEEP(hn[c]); }
    if (!(p.dbg & 1)) stage_unit(gw, h0);
  }
  bool firstu = true;
  for (int u = gw; u < p.nunits; u += nw) {
    const int row0 = p.unit[4 * u], nown = p.unit[4 * u + 1];
    // this unit's stage has landed (requested before the previous unit's four stores), and so have its rulebook and the next unit's halo indices
    if (firstu) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (!firstu) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) { KEEP(rbn[t][q]); rbc[t][q] = rbn[t][q]; }
#pragma unroll
      for (int c = 0; c < W_HCH; ++c) KEEP(hn[c]);
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) KEEP(rbc[t][q]);
    }
    firstu = false;
    int hcur[W_HCH];
#pragma unroll
    for (int c = 0; c < W_HCH; ++c) hcur[c] = hn[c];                                  // indices of unit t + 1 (for the staging after the taps)
    load_hidx(u + 2 * nw, hn);                                                       // unit t + 2
    load_rb(u + nw, rbn);                                                            // unit t + 1

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    if (!(p.dbg & 2)) {
      constexpr int D = 2;                                                           // 6 reads per tap: lgkmcnt counts to 15
      u32x4 A[D][2][2], B[D][2];
      auto issue = [&](int k, int s_) __attribute__((always_inline)) {
        B[s_][0] = lds_r128(boff[0] + (unsigned)(k * 2048));
        B[s_][1] = lds_r128(boff[1] + (unsigned)(k * 2048));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint32_t wd = rbc[t][k >> 3][(k >> 1) & 3];
          const unsigned val = (k & 1) ? (wd >> 16) : (wd & 0xFFFFu);
          const unsigned a0 = st_a + val;
          A[s_][t][0] = lds_r128(a0 ^ pc16[0]);
          A[s_][t][1] = lds_r128(a0 ^ pc16[1]);
        }
      };
      issue(0, 0);
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const int s_ = k & 1;
        if (k + 1 < 27) { issue(k + 1, s_ ^ 1); LGKM(6); } else LGKM(0);
        KEEP(B[s_][0]); KEEP(B[s_][1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          KEEP(A[s_][t][0]); KEEP(A[s_][t][1]);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[s_][t][0]), __builtin_bit_cast(bf16x8, B[s_][0]), acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[s_][t][1]), __builtin_bit_cast(bf16x8, B[s_][1]), acc[t], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* ew = reinterpret_cast<float*>(stage);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) ew[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 36 + fi] = acc[t][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32x4 o[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = (lane >> 2) + 16 * it, cvv = lane & 3;
      const unsigned ea = st_a + (unsigned)((rr * 36 + cvv * 8) * 4);
      u32x4 u0 = lds_r128(ea), u1 = lds_r128(ea + 16);
      LGKM(0);
      KEEP(u0); KEEP(u1);
      o[it] = u32x4{pack2(__uint_as_float(u0[0]), __uint_as_float(u0[1])), pack2(__uint_as_float(u0[2]), __uint_as_float(u0[3])),
                    pack2(__uint_as_float(u1[0]), __uint_as_float(u1[1])), pack2(__uint_as_float(u1[2]), __uint_as_float(u1[3]))};
    }
    // the stage is free (the results are in registers): the next unit's staging goes out BEFORE this unit's stores, so that the wait at the
    // top of the next iteration can leave the stores outstanding
    if (!(p.dbg & 1)) stage_unit(u + nw, hcur);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = (lane >> 2) + 16 * it, cvv = lane & 3;
      const unsigned off = (rr < nown && !(p.dbg & 4)) ? (unsigned)(row0 + rr) * 64u + (unsigned)(cvv * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o[it]), ro, (int)off, 0, 0);
    }
  }
}

extern "C" int conv_blk_w(const void* x, const void* w, void* out, const void* unit, const void* halo, const void* lrb, int64_t n, int nunits, int waves,
                          int dbg, void* stream) {
  BlkS p{(const uint16_t*)x, (const uint16_t*)w, (uint16_t*)out, (const int32_t*)unit, (const int32_t*)halo, (const uint16_t*)lrb, n, nunits, dbg};
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)WS_B + (size_t)waves * W_STAGE_B;
  if (lds > 160 * 1024) return -4;
#define GOW(WW)                                                                                                                  \
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_blk_w<WW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; \
  k_conv_blk_w<WW><<<256, WW * 64, lds, s>>>(p);
  if (waves == 8) { GOW(8) } else if (waves == 6) { GOW(6) } else if (waves == 4) { GOW(4) } else return -2;
#undef GOW
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int conv_blk(const void* x, const void* w, void* out, const void* unit, const void* halo, const void* lrb, int64_t n, int nunits, int waves,
                        int dbg, void* stream) {
  BlkP p{(const uint16_t*)x, (const uint16_t*)w, (uint16_t*)out, (const int32_t*)unit, (const int32_t*)halo, (const uint8_t*)lrb, n, nunits, dbg};
  hipStream_t s = (hipStream_t)stream;
#define GO(W_)                                                                                                                   \
  {                                                                                                                              \
    const size_t lds = WS_B + (size_t)W_ * STAGE_B;                                                                              \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_blk<W_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; \
    k_conv_blk<W_><<<256, W_ * 64, lds, s>>>(p);                                                                                 \
  }
  if (waves == 6) GO(6) else if (waves == 4) GO(4) else if (waves == 5) GO(5)
  else if (waves == 104 || waves == 102 || waves == 101) {               // variant R: W = waves - 100 waves per workgroup, weights in registers
    const int W_ = waves - 100;
    const size_t lds = (size_t)W_ * 2 * STAGE_B;
    const int per_cu = 4 / W_;                                           // one wave per SIMD
#define GOR(WW)                                                                                                                  \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_blk_r<WW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; \
    k_conv_blk_r<WW><<<256 * per_cu, WW * 64, lds, s>>>(p);
    if (W_ == 4) { GOR(4) } else if (W_ == 2) { GOR(2) } else { GOR(1) }
#undef GOR
  } else return -2;
#undef GO
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
