// PROTOTYPE v2 (developer tool, not part of the library): conv_l2.hip with staging, descriptor loads and the epilogue OVERLAPPED with the
// tap loop -- what DESIGN.md R4.9 (ii) asks for.
//
// Same unit as v1 (384 rows of the block-local order of level 2, 4 waves x 96 rows, <= 1 023 stage positions), but the 64 input channels are
// contracted in two phases of 32: two half-row stages of 64 KB (1 024 positions x 64 B).  While the taps of phase h run out of stage h,
// the other stage is filled by LDS-DMA for the next phase (phase 1 of this unit / phase 0 of the next unit), the next unit's halo row ids
// and rulebook rows are requested into registers, and nothing but two barriers per unit separates the phases.  Every tap issues its 4
// weight-fragment loads LA taps ahead plus exactly 3 "other" loads (DMA pieces, descriptor words, or an out-of-range dummy); the counted
// wait in front of tap K is a compile-time constant (taps are template instances).
//
// RESULT (profiles/r4_final/proto_l2.txt): correct, and SLOWER than v1 -- 0.37-0.42 ms (v1 0.334, gather kernel 0.28); the first build ran at
// the full 2.43 GHz with 26 % MFMA-busy (rocprofv3 PMC, tools/proto_l2/pmc.sh), i.e. stalled, not power-limited.  `vmcnt` retires loads in order: a DMA piece is an HBM round trip, and the weights of tap
// K + LA requested behind it cannot be waited for before it has landed, so with LA = 2 every piece stalls its wave; LA = 4 / 6 need 80 / 112
// registers of weight fragments and spill (0.497 ms).  Overlapping the staging with the taps of the SAME waves needs the weights off the
// vmcnt path (an LDS ring filled by a loader wave) or a loader wave for the stage -- the next thing to build, not a tuning of this file.
//
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/proto_l2/conv_l2_v2.hip -o tools/proto_l2/libproto_l2_v2.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;

namespace {

constexpr int U = 384, WAVES = 4, NT = WAVES * 64, MT = 3;
constexpr int POS = 1024, ZERO = 1023;
constexpr int HMAX = POS - 1 - U;
constexpr int SB = POS * 64;                             // bytes of one half-row stage
constexpr int NQ = POS / 16 / WAVES;                     // DMA instructions per wave and stage (16): one = 16 positions x 64 B
constexpr int OTH = 3;                                   // "other" loads per tap
constexpr int LA = 2;                                    // taps of lookahead of the weight fragments
// loads issued after the weights of tap k were requested and before tap k starts (what may still be in flight when they are needed)
constexpr int younger(int k) {
  int c = 0;
  if (k < LA) { c += 4 * (LA - 1 - k); for (int t = 0; t < k; ++t) c += (t + LA < 27 ? 4 : 0) + OTH; return c; }   // requested in the phase's preamble
  c += OTH;                                               // the rest of tap k - LA
  for (int t = k - LA + 1; t < k; ++t) c += (t + LA < 27 ? 4 : 0) + OTH;
  return c;
}


struct P {
  const uint16_t* x; const uint16_t* wfrag; uint16_t* out;
  const int32_t* halo; const int32_t* nhalo; const uint32_t* lrb;
  int64_t n; int units;
};

__device__ __forceinline__ u32x4 lds_r128(unsigned a) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); return v; }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { const bf16x2 v = {(__bf16)lo, (__bf16)hi}; return __builtin_bit_cast(uint32_t, v); }
// 16-B piece pc (0..3) of stage position pos in a 64-B row
__device__ __forceinline__ unsigned st_off(unsigned pos, unsigned pc) { return pos * 64u + ((pc ^ ((pos >> 1) & 3u)) << 4); }

__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) k_conv_l2v2(P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];            // [2][POS][64 B] + [WAVES][32][36] fp32
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wfrag), 0, 27 * 64 * 64 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, (int)(p.n * 128), 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.halo), 0, (int)((int64_t)p.units * HMAX * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.lrb), 0, (int)(p.n * 36), 0x00020000);
  float* ew = reinterpret_cast<float*>(smem + 2 * SB) + wv * 32 * 36;

  // the stage position the lane fills in DMA instruction i of its wave, and the halo slot behind it (-1: an own row)
  auto q_pos = [&](int i) __attribute__((always_inline)) { return ((wv + WAVES * i) * 16) + (lane >> 2); };
  // ---- pieces of work that ride on the taps as "other" loads
  auto load_hrow = [&](int u, int i) __attribute__((always_inline)) -> int {       // halo row id behind DMA instruction i (0 for own rows; < 0: nothing)
    const int pos = q_pos(i);
    const unsigned off = (pos >= U && pos < POS - 1 && u < p.units) ? ((unsigned)u * HMAX + (unsigned)(pos - U)) * 4u : 0xFFFFFFFFu;
    return __builtin_amdgcn_raw_buffer_load_b32(rh, (int)off, 0, 0);               // out of range reads 0
  };
  auto dma = [&](int u, int h, int i, int hr) __attribute__((always_inline)) {     // 16 positions of stage h of unit u
    const int pos = q_pos(i);
    int64_t r = pos < U ? (int64_t)u * U + pos : (int64_t)hr;
    // halo slots past the unit's halo hold -1 in the list; own rows past the end of the tensor and the zero row fall out of the buffer
    const bool ok = u < p.units && pos < POS - 1 && r >= 0 && r < p.n && (pos < U || hr >= 0);
    const unsigned off = ok ? (unsigned)r * 128u + (unsigned)(h * 64) + (unsigned)((((lane & 3) ^ ((pos >> 1) & 3))) * 16) : 0xFFFFFFFFu;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + h * SB + (wv + WAVES * i) * 1024), 16, (int)off, 0, 0, 0);
  };
  auto load_rb = [&](int u, int t, int q) __attribute__((always_inline)) -> uint32_t {
    const int64_t r = (int64_t)u * U + wv * (32 * MT) + t * 32 + fi;
    const unsigned off = (u < p.units && r < p.n) ? (unsigned)(r * 36 + q * 4) : 0xFFFFFFFFu;
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rl, (int)off, 0, 0);
  };
  auto dummy = [&]() __attribute__((always_inline)) { int v = __builtin_amdgcn_raw_buffer_load_b32(rh, (int)0xFFFFFFFFu, 0, 0); asm volatile("" ::"v"(v)); };

  int hrow[NQ], hrown[NQ];
  uint32_t rb[MT][9], rbn[MT][9];
  int u = blockIdx.x;
  // ---- prologue: the first unit's descriptors and its phase-0 stage, synchronously
#pragma unroll
  for (int i = 0; i < NQ; ++i) hrow[i] = load_hrow(u, i);
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int q = 0; q < 9; ++q) rb[t][q] = load_rb(u, t, q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NQ; ++i) { asm volatile("" : "+v"(hrow[i])); hrow[i] = (q_pos(i) >= U) ? hrow[i] - 0 : 0; }
#pragma unroll
  for (int i = 0; i < NQ; ++i) dma(u, 0, i, hrow[i]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (; u < p.units; u += gridDim.x) {
    const int un = u + (int)gridDim.x;
    f32x16 acc[MT][2];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][nb][i] = 0.f;

#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned sbase = lds0 + (unsigned)(h * SB);
      u32x4 A[2][MT][2], B[LA + 1][2][2];
      auto issue_b = [&](int k, int b_) __attribute__((always_inline)) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int j = 0; j < 2; ++j)      // fragment order: vector ((((k * 2 + nb) * 2 + h) * 2 + j) * 64 + lane)
            B[b_][nb][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, ((((k * 2 + nb) * 2 + h) * 2 + j) * 64) * 16, 0));
      };
      auto issue_a = [&](int k, int a_) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const unsigned pos = (rb[t][k / 3] >> (10 * (k % 3))) & 1023u;
#pragma unroll
          for (int j = 0; j < 2; ++j) A[a_][t][j] = lds_r128(sbase + st_off(pos, (unsigned)(2 * j + fh)));
        }
      };
      // the "other" load number o (0 .. 27 * OTH - 1) of this phase
      auto other = [&](int o) __attribute__((always_inline)) {
        if (o < NQ) {                                     // fill the other stage: phase 1 of this unit / phase 0 of the next one
          if (h == 0) dma(u, 1, o, hrow[o]); else dma(un, 0, o, hrown[o]);
        } else if (h == 0 && o < 2 * NQ) hrown[o - NQ] = load_hrow(un, o - NQ);      // next unit's halo ids (needed by phase 1's DMA)
        else if (h == 1 && o < NQ + MT * 9) { const int e = o - NQ; rbn[e / 9][e % 9] = load_rb(un, e / 9, e % 9); }
        else dummy();
      };
#pragma unroll
      for (int k = 0; k < LA; ++k) issue_b(k, k);
      issue_a(0, 0);
      auto tap = [&]<int K>() __attribute__((always_inline)) {
        constexpr int b_ = K % (LA + 1), a_ = K & 1;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger(K)) : "memory");
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(B[b_][nb][j]));
        if constexpr (K + LA < 27) issue_b(K + LA, (K + LA) % (LA + 1));
#pragma unroll
        for (int o = 0; o < OTH; ++o) other(K * OTH + o);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (K + 1 < 27) { issue_a(K + 1, a_ ^ 1); asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * MT) : "memory"); } else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(A[a_][t][j]));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
              acc[t][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[a_][t][j]), __builtin_bit_cast(bf16x8, B[b_][nb][j]), acc[t][nb], 0, 0, 0);
      };
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) { (tap.template operator()<Ks>(), ...); }(std::make_integer_sequence<int, 27>{});
      if (h == 1) {
        // ---- epilogue through the wave's own [32][36] tile (the stages stay untouched: stage 0 is being filled for the next unit)
        const int64_t row0 = (int64_t)u * U;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ew[((r & 3) + 8 * (r >> 2) + 4 * fh) * 36 + fi] = acc[t][nb][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e0 = 0; e0 < 2; ++e0) {
              const int rr = (lane >> 2) + 16 * e0, cv = lane & 3;
              const int64_t r = row0 + wv * (32 * MT) + t * 32 + rr;
              if (r < p.n) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(ew + rr * 36 + cv * 8), v1 = *reinterpret_cast<const f32x4*>(ew + rr * 36 + cv * 8 + 4);
                const u32x4 o = {pack2(v0[0], v0[1]), pack2(v0[2], v0[3]), pack2(v1[0], v1[1]), pack2(v1[2], v1[3])};
                *reinterpret_cast<u32x4*>(p.out + r * 64 + nb * 32 + cv * 8) = o;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's DMA pieces of the other stage have landed
      if (h == 0) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) asm volatile("" : "+v"(hrown[i]));
      } else {
#pragma unroll
        for (int i = 0; i < NQ; ++i) hrow[i] = hrown[i];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int q = 0; q < 9; ++q) { asm volatile("" : "+v"(rbn[t][q])); rb[t][q] = rbn[t][q]; }
      }
      __syncthreads();                                                   // everyone is done reading stage h; the other stage is complete
    }
  }
}

}  // namespace

extern "C" int proto_l2v2_conv(const void* x, const void* wfrag, void* out, const int32_t* halo, const int32_t* nhalo, const uint32_t* lrb,
                               int64_t n, int units, void* stream) {
  P p{(const uint16_t*)x, (const uint16_t*)wfrag, (uint16_t*)out, halo, nhalo, lrb, n, units};
  const int lds = 2 * SB + WAVES * 32 * 36 * 4;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_l2v2), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
    attr = true;
  }
  const int grid = units < 256 ? units : 256;
  k_conv_l2v2<<<grid, NT, lds, (hipStream_t)stream>>>(p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
