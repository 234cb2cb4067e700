// Micro-benchmark: what a gfx950 part sustains for the STORE patterns the conv epilogues use.  A wave owns 32 consecutive rows of a
// [n][C] bf16 matrix and writes them with 16-B stores per lane in one of these shapes:
//   seg64 : per instruction 16 rows x 64 B (4 lanes per row), the other 64-B halves of the rows by the NEXT instruction
//           (the stream / streamq kernels' epilogue: one 32-column block at a time)
//   row128: per instruction 8 whole 128-B rows (8 lanes per row)
//   seg64x2 / row128x2: the same into TWO matrices (a second output view)
// with 4 or 8 waves per workgroup, one tile per wave (the conv kernels' shape: stores arrive in bursts at the end of a workgroup's life).
//   hipcc --offload-arch=gfx950 -O3 tools/store_roof.hip -o /tmp/store_roof && /tmp/store_roof
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// MODE 0: seg64, 1: row128;  NV views;  rows of C = 64 bf16 (128 B), leading dimension ld elements
template <int MODE, int NV>
__global__ void __launch_bounds__(512) k_store(uint16_t* __restrict__ o0, uint16_t* __restrict__ o1, int64_t n, int64_t ld, uint32_t seed) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t r0 = ((int64_t)blockIdx.x * nw + wv) * 32;
  const u32x4 v = {seed + lane, seed ^ (uint32_t)r0, seed * 3u, seed + 7u};
  uint16_t* outs[2] = {o0, o1};
  if (MODE == 0) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e0 = 0; e0 < 2; ++e0) {
        const int e = lane + e0 * 64, rr = e >> 2, cv = e & 3;
        if (r0 + rr < n) {
#pragma unroll
          for (int q = 0; q < NV; ++q) *reinterpret_cast<u32x4*>(outs[q] + (r0 + rr) * ld + nb * 32 + cv * 8) = v;
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = i * 8 + (lane >> 3), cv = lane & 7;
      if (r0 + rr < n) {
#pragma unroll
        for (int q = 0; q < NV; ++q) *reinterpret_cast<u32x4*>(outs[q] + (r0 + rr) * ld + cv * 8) = v;
      }
    }
  }
}

template <int MODE, int NV>
static int run(const char* name, uint16_t* a, uint16_t* b, int64_t n, int64_t ld, int waves) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((n + 32 * waves - 1) / (32 * waves));
  for (int i = 0; i < 3; ++i) k_store<MODE, NV><<<grid, waves * 64>>>(a, b, n, ld, i);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k_store<MODE, NV><<<grid, waves * 64>>>(a, b, n, ld, i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double gb = (double)n * 128 * NV / 1e9;
  printf("%-10s views %d  ld %3lld  waves/wg %d  %7.3f ms  %6.0f GB/s\n", name, NV, (long long)ld, waves, ms, gb / ms * 1e3);
  return 0;
}

int main() {
  const int64_t n = 1105126;                       // level 2 of the config-2 tile
  uint16_t *a, *b;
  CK(hipMalloc(&a, n * 128 * 2)); CK(hipMalloc(&b, n * 128 * 2));
  for (int waves : {4, 8})
    for (int64_t ld : {64, 128}) {
      if (run<0, 1>("seg64", a, b, n, ld, waves)) return 1;
      if (run<1, 1>("row128", a, b, n, ld, waves)) return 1;
      if (run<0, 2>("seg64", a, b, n, ld, waves)) return 1;
      if (run<1, 2>("row128", a, b, n, ld, waves)) return 1;
    }
  return 0;
}
