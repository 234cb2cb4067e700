// PROTOTYPE v3 (developer tool, not part of the library): the staged-unit level-2 conv with LOADER waves.
//
// v2 showed that the compute waves cannot also issue the staging DMAs: `vmcnt` retires in order and couples their weight-fragment loads to
// the HBM round trips of the DMA pieces.  Here a workgroup has 4 compute waves (one per SIMD, 96 rows x 64 output channels each, as in v1)
// and 4 loader waves (one per SIMD beside them, a handful of registers): the loaders fill the OTHER half-row stage (and fetch the next
// unit's halo row ids) while the compute waves run the 27 taps of the current phase; two workgroup barriers per unit hand the stages over.
// Compute waves only ever wait for their own weight fragments (two taps ahead) and, once per unit, their rulebook rows; their accumulator
// tiles are formed transposed (weights as the MFMA's row operand) so the epilogue stores 8-B pieces straight from registers.
//
// RESULT (profiles/r4_final/proto_l2.txt): correct (bit-identical to v2), 0.30-0.31 ms against 0.26-0.28 for the gather kernel with one view;
// 0.20-0.21 with neither DMAs nor stores, 0.24-0.26 with one of them: the bytes cost their time although nothing waits for them.
//
//   hipcc -std=c++20 --offload-arch=gfx950 -O3 -shared -fPIC tools/proto_l2/conv_l2_v3.hip -o tools/proto_l2/libproto_l2_v3.so
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

constexpr int U = 384, CW = 4, LW = 4, NT = (CW + LW) * 64, MT = 3;
constexpr int POS = 1024;
constexpr int HMAX = POS - 1 - U;
constexpr int SB = POS * 64;                             // bytes of one half-row stage
constexpr int NQ = POS / 16 / LW;                        // DMA instructions per loader wave and stage (16): one = 16 positions x 64 B

struct P {
  const uint16_t* x; const uint16_t* wfrag; uint16_t* out;
  const int32_t* halo; const int32_t* nhalo; const uint32_t* lrb;
  int64_t n; int units; int mode;      // mode (timing ablations): 1 the loaders issue no DMA, 4 no output stores
};

__device__ __forceinline__ u32x4 lds_r128(unsigned a) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); return v; }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { const bf16x2 v = {(__bf16)lo, (__bf16)hi}; return __builtin_bit_cast(uint32_t, v); }
// 16-B piece pc (0..3) of stage position pos in a 64-B row
__device__ __forceinline__ unsigned st_off(unsigned pos, unsigned pc) { return pos * 64u + ((pc ^ ((pos >> 1) & 3u)) << 4); }

__global__ void __launch_bounds__(NT) k_conv_l2v3(P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];            // [2][POS][64 B] + [CW][32][36] fp32
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int G = (int)gridDim.x;

  if (wv >= CW) {
    // ================================================================ loader waves
    const int lw = wv - CW;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, (int)(p.n * 128), 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.halo), 0, (int)((int64_t)p.units * HMAX * 4), 0x00020000);
    auto q_pos = [&](int i) __attribute__((always_inline)) { return ((lw + LW * i) * 16) + (lane >> 2); };
    auto load_hrow = [&](int u, int (&h)[NQ]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int pos = q_pos(i);
        const unsigned off = (pos >= U && pos < POS - 1 && u < p.units) ? ((unsigned)u * HMAX + (unsigned)(pos - U)) * 4u : 0xFFFFFFFFu;
        h[i] = __builtin_amdgcn_raw_buffer_load_b32(rh, (int)off, 0, 0);
      }
    };
    auto fill = [&](int u, int h, const int (&hr)[NQ]) __attribute__((always_inline)) {      // stage h of unit u: 16 positions per instruction
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int pos = q_pos(i);
        const int64_t r = pos < U ? (int64_t)u * U + pos : (int64_t)hr[i];
        const bool ok = u < p.units && pos < POS - 1 && r >= 0 && r < p.n && (pos < U || hr[i] >= 0);
        const unsigned off = ok ? (unsigned)r * 128u + (unsigned)(h * 64) + (unsigned)((((lane & 3) ^ ((pos >> 1) & 3))) * 16) : 0xFFFFFFFFu;
        if (!(p.mode & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + h * SB + (lw + LW * i) * 1024), 16, (int)off, 0, 0, 0);
      }
    };
    int hrow[NQ], hrown[NQ];
    int u = blockIdx.x;
    load_hrow(u, hrow);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NQ; ++i) asm volatile("" : "+v"(hrow[i]));
    fill(u, 0, hrow);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (; u < p.units; u += G) {
      const int un = u + G;
      fill(u, 1, hrow);                                   // under the taps of phase 0
      load_hrow(un, hrown);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NQ; ++i) asm volatile("" : "+v"(hrown[i]));
      __syncthreads();
      fill(un, 0, hrown);                                 // under the taps of phase 1 and the epilogue
#pragma unroll
      for (int i = 0; i < NQ; ++i) hrow[i] = hrown[i];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    return;
  }

  // ================================================================== compute waves
  const int fi = lane & 31, fh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wfrag), 0, 27 * 64 * 64 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.lrb), 0, (int)(p.n * 36), 0x00020000);
  float* ew = reinterpret_cast<float*>(smem + 2 * SB) + wv * 32 * 36;
  uint32_t rb[MT][9];
  auto load_rb = [&](int u) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int64_t r = (int64_t)u * U + wv * (32 * MT) + t * 32 + fi;
#pragma unroll
      for (int q = 0; q < 9; ++q) rb[t][q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rl, (int)((u < p.units && r < p.n) ? (unsigned)(r * 36 + q * 4) : 0xFFFFFFFFu), 0, 0);
    }
  };
  int u = blockIdx.x;
  load_rb(u);
  __syncthreads();
  for (; u < p.units; u += G) {
    const int un = u + G;
    f32x16 acc[MT][2];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][nb][i] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the unit's rulebook rows (requested during the previous epilogue)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int q = 0; q < 9; ++q) asm volatile("" : "+v"(rb[t][q]));
#pragma nounroll
    for (int h = 0; h < 2; ++h) {
      const unsigned sbase = lds0 + (unsigned)(h * SB);
      u32x4 A[2][MT][2], B[3][2][2];
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
      issue_b(0, 0); issue_b(1, 1);
      issue_a(0, 0);
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const int b_ = k % 3, a_ = k & 1;
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < 27) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // B(k); B(k + 1) may be in flight
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(B[b_][nb][j]));
        if (k + 2 < 27) issue_b(k + 2, (k + 2) % 3);
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < 27) { issue_a(k + 1, a_ ^ 1); asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * MT) : "memory"); } else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
              acc[t][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, B[b_][nb][j]), __builtin_bit_cast(bf16x8, A[a_][t][j]), acc[t][nb], 0, 0, 0);   // (weights as the row operand: the tile comes out transposed)
      }
      if (h == 1) {
        load_rb(un);                                                     // the next unit's rulebook rows, under the epilogue
        // the accumulator tiles are TRANSPOSED (output channels down the tile, rows across the lanes): lane (fi, fh) holds, for ITS row, the
        // channels 8 q + 4 fh .. + 4 of the block (q = 0..3) -- four consecutive channels per register quad, stored straight from registers
        // as 8-B pieces; no LDS transposition, no wave barriers
        const int64_t row0 = (int64_t)u * U;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int64_t r = row0 + wv * (32 * MT) + t * 32 + fi;
          if (r < p.n && !(p.mode & 4)) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const uint2 o = make_uint2(pack2(acc[t][nb][4 * q], acc[t][nb][4 * q + 1]), pack2(acc[t][nb][4 * q + 2], acc[t][nb][4 * q + 3]));
                *reinterpret_cast<uint2*>(p.out + r * 64 + nb * 32 + 8 * q + 4 * fh) = o;
              }
          }
        }
      }
      __syncthreads();                                                   // stage h is free for the loaders; the other stage is complete
    }
  }
}

}  // namespace

extern "C" int proto_l2v3_conv(const void* x, const void* wfrag, void* out, const int32_t* halo, const int32_t* nhalo, const uint32_t* lrb,
                               int64_t n, int units, void* stream, int mode) {
  P p{(const uint16_t*)x, (const uint16_t*)wfrag, (uint16_t*)out, halo, nhalo, lrb, n, units, mode};
  const int lds = 2 * SB + CW * 32 * 36 * 4;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_l2v3), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
    attr = true;
  }
  const int grid = units < 256 ? units : 256;
  k_conv_l2v3<<<grid, NT, lds, (hipStream_t)stream>>>(p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
