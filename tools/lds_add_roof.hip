// What does the scatter-add of a "gather-GEMM-scatter per rule" formulation cost on this part?  (level 2, present pairs only.)
//
// A per-tap-compacted kernel multiplies 32 COMPACTED rows of a tap by W[tap] (8 MFMAs 32x32x16 for 64 -> 64) and must then add the 32 x 64
// fp32 product tile into the accumulator rows of a unit held in LDS, because the compacted rows differ from tap to tap: 32 ds_add_f32 per
// lane and (tile, tap) -- lanes 0..31 one output row, lanes 32..63 another (the C layout of the 32x32 MFMA), i.e. the friendliest possible
// bank pattern.  This program times, on every CU at once (W waves per CU):
//   add   : 32 ds_add_f32 (no return) per (tile, tap), rows drawn at random inside a 256-row unit tile (64 KB of fp32 accumulators)
//   write : the same addresses with ds_write_b32 (the rate MI355X_MICROARCH.md quotes: 64 B/clk/CU)
//   mfma  : the 8 MFMAs of that (tile, tap), for scale
//   both  : MFMAs and adds interleaved (do they overlap?)
// and prints cycles per (tile, tap) per CU at the measured clock.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_add_roof.hip -o tools/lds_add_roof && ./tools/lds_add_roof
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(512) k(const int* __restrict__ rows, int iters, float* __restrict__ sink) {
  extern __shared__ float acc[];                                   // [256][64] fp32 = 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int e = tid; e < 256 * 64; e += blockDim.x) acc[e] = 0.f;
  __syncthreads();
  f32x16 c[2];
  for (int i = 0; i < 16; ++i) { c[0][i] = 0.f; c[1][i] = 0.f; }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  const int salt = rows[(blockIdx.x * 8 + wv) & 1023];            // (one load per wave, outside the loop: the row pattern is arithmetic)
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2 || MODE == 3) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[0], 0, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[1], 0, 0, 0);
      }
    }
    if (MODE != 2) {
      // C layout: register g of lane (n = lane & 31, h = lane >> 5) is row (g & 3) + 8 (g >> 2) + 4 h of the compacted tile
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int m = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5);
          const int orow = (it * 37 + salt + m * 5) & 255;          // 32 distinct output rows of the unit per compacted tile, 2 per instruction
          float* dst = acc + orow * 64 + nb * 32 + (lane & 31);
          const float v = (MODE == 3) ? c[nb][g] : 1.0f;
          if (MODE == 1) *(volatile float*)dst = v;
          else __hip_atomic_fetch_add(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
  }
  __syncthreads();
  float s = c[0][0] + c[1][3];
  for (int e = tid; e < 256 * 64; e += blockDim.x) s += acc[e];
  if (s == 123.456f) sink[0] = s;
}

template <int MODE>
static float run(const int* rows, int waves, int iters, float* sink) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, waves * 64, 64 * 1024>>>(rows, iters, sink);
  hipEventRecord(e0);
  k<MODE><<<256, waves * 64, 64 * 1024>>>(rows, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  std::vector<int> h(256 * 8 * 64 * 32);
  srand(1);
  for (size_t i = 0; i < h.size(); i += 32) {                      // 32 DISTINCT rows of the 256-row unit per compacted tile
    int base = rand() % 224;
    for (int j = 0; j < 32; ++j) h[i + j] = base + ((j * 7 + rand() % 3) % 32);
    for (int j = 0; j < 32; ++j) h[i + j] = base + ((h[i + j] - base) % 32);
  }
  int* rows; float* sink;
  hipMalloc(&rows, h.size() * 4); hipMalloc(&sink, 4);
  hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 4096;
  const char* names[4] = {"add   (32 ds_add_f32)", "write (32 ds_write_b32)", "mfma  (8 x 32x32x16)", "both  (8 MFMAs + 32 ds_add_f32 of their results)"};
  for (int waves : {4, 8}) {
    printf("%d waves per CU, %d (tile, tap) items per wave\n", waves, iters);
    float ms[4] = {run<0>(rows, waves, iters, sink), run<1>(rows, waves, iters, sink), run<2>(rows, waves, iters, sink), run<3>(rows, waves, iters, sink)};
    // clock: the MFMA run is 8 x 32 cycles per item per SIMD, waves / 4 waves per SIMD
    const double mfma_cycles = 8.0 * 32.0 * iters * (waves / 4.0);
    const double ghz = mfma_cycles / (ms[2] * 1e-3) / 1e9;
    for (int m = 0; m < 4; ++m)                                     // CU time per item: the four SIMDs of a CU work on four items at once
      printf("  %-52s %8.3f ms = %7.1f cycles of CU time per (tile, tap) item at the MFMA run's %.2f GHz (= %.1f x the item's 8 MFMAs)\n", names[m], ms[m],
             ms[m] * 1e-3 * ghz * 1e9 / iters / waves, ghz, ms[m] / ms[2]);
  }
  printf("reading: a present-pairs kernel at level 2 saves 0.40 of the dense form's MFMAs (26 of 64 cycles of CU time per dense (tile, tap)) and pays the\n"
         "`add` line for 0.60 of them -- it loses as soon as `add` exceeds 43 cycles of CU time per item (0.67 x the item's MFMAs).\n");
  return 0;
}
