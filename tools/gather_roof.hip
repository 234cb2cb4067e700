// Micro-benchmark: how fast can a gfx950 CU GATHER rows (16 B per lane, `buffer_load_dwordx4` on a buffer resource) -- the access every
// sparse-conv kernel of this repo is made of (one rulebook entry -> one feature row of 64 .. 256 B).  Measures the machine's gather roof so
// that DESIGN.md can price the conv / wgrad kernels against it instead of against the streaming HBM figure.
//
//   hipcc --offload-arch=gfx950 -O3 tools/gather_roof.hip -o /tmp/gather_roof && /tmp/gather_roof
//
// Per configuration: row size, window the row indices are drawn from (per XCD: L2-resident, MALL-resident, HBM), share of present rows (absent
// = out-of-range offset, returns zeros without a memory access -- the rulebook's -1), loads in flight per wave, waves per CU, destination
// (registers or LDS-DMA).  Prints requested bytes (present lanes only) per second, and lanes (present or not) per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// One wave gathers `iters` batches of UNR instructions; lane l of an instruction fetches piece l % LPR of row r(instruction, l / LPR).
// sequential != 0: the rows of a wave are consecutive (a streaming read through the same instruction) -- the control.
template <int ROWB, int UNR, bool DMA, int LPR = ROWB / 16, bool EXECMASK = false, bool SPLIT = false>
__global__ void __launch_bounds__(1024) k_gather(const uint32_t* __restrict__ base, uint32_t bytes, uint32_t win_rows, uint32_t present_1024, int iters,
                                                 int sequential, uint32_t* __restrict__ sink) {
  constexpr int RPI = 64 / LPR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(base), 0, (int)bytes, 0x00020000);
  const uint32_t nrows = bytes / ROWB;
  const uint32_t xcd = blockIdx.x & 7;                                  // workgroup b runs on XCD b % 8: every XCD draws from its own window
  const uint32_t w0 = (uint32_t)(((uint64_t)xcd * (nrows - win_rows)) / 8);
  const uint32_t gw = (blockIdx.x * nw + wv);
  u32x4 acc = {0, 0, 0, 0};
  uint32_t seq = mix(gw) % win_rows;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const uint32_t rslot = SPLIT ? (uint32_t)(lane % RPI) : (uint32_t)(lane / LPR), piece = SPLIT ? (uint32_t)(lane / RPI) : (uint32_t)(lane % LPR);
      const uint32_t h = mix((gw * 9176u + (uint32_t)it) * 64u + (uint32_t)u * 64u + rslot);
      uint32_t row = sequential ? (seq + rslot) % win_rows : h % win_rows;
      seq += sequential ? RPI : 0;
      const bool here = (mix(h ^ 0x9e3779b9u) & 1023u) < present_1024;
      const uint32_t off = here ? (w0 + row) * ROWB + piece * 16u : 0xFFFFFFFFu;
      if constexpr (DMA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (wv * UNR + u) * 1024), 16, (int)off, 0, 0, 0);
      else if constexpr (EXECMASK) { v[u] = u32x4{0, 0, 0, 0}; if (here) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0)); }
      else v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
    }
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc ^= v[u];
    }
  }
  if constexpr (DMA) acc[0] = *reinterpret_cast<uint32_t*>(smem + threadIdx.x * 4);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;   // keeps the loads alive
}

template <int ROWB, int UNR, bool DMA, int LPR = ROWB / 16, bool EXECMASK = false, bool SPLIT = false>
static int run(const uint32_t* buf, size_t bytes, size_t win_bytes, int present_pct, int waves_per_cu, int sequential, uint32_t* sink, double clk_ghz) {
  const int cus = 256, iters = 2000 / UNR * 4;
  const int wpb = waves_per_cu >= 16 ? 16 : waves_per_cu;                 // one workgroup per CU up to 16 waves, two beyond
  const int blocks = cus * (waves_per_cu / wpb);
  const uint32_t win_rows = (uint32_t)(win_bytes / ROWB);
  const uint32_t pres = (uint32_t)(present_pct * 1024 / 100);
  const size_t lds = DMA ? (size_t)wpb * UNR * 1024 : 0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    if (rep) CK(hipEventRecord(e0));
    k_gather<ROWB, UNR, DMA, LPR, EXECMASK, SPLIT><<<blocks, wpb * 64, lds>>>(buf, (uint32_t)bytes, win_rows, pres, iters, sequential, sink);
    if (rep) CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
  }
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double instr = (double)blocks * wpb * iters * UNR;
  const double req = instr * 1024.0 * present_pct / 100.0;   // 16 B per present lane
  printf("row %3d B x %2d lanes%s  window/XCD %7.1f MB  present %3d %%  %2d waves/CU x %d in flight  %s %s: %7.2f TB/s requested, %5.1f lanes/clk/CU at %.1f GHz (%.3f ms)\n",
         ROWB, LPR, EXECMASK ? " (absent lanes EXEC-masked)" : SPLIT ? " (a row's lanes 64/LPR apart: MFMA-fragment shape)" : "", win_bytes / 1048576.0, present_pct, waves_per_cu, UNR, DMA ? "LDS-DMA  " : "registers", sequential ? "sequential" : "random    ",
         req / (ms * 1e-3) / 1e12, instr * 64.0 / (ms * 1e-3) / (clk_ghz * 1e9) / cus, clk_ghz, ms);
  return 0;
}

int main() {
  const size_t bytes = (size_t)1800 << 20;                                // < 2^31: 32-bit buffer offsets
  uint32_t *buf, *sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, bytes)); CK(hipMemset(sink, 0, 64));
  const double ghz = 2.4;
  const size_t MB = 1 << 20;
  // 1. the roof by residency (128-B rows = a 64-channel bf16 row, all present, 16 waves x 8 in flight)
  for (size_t win : {1 * MB, 2 * MB, 16 * MB, 200 * MB}) run<128, 8, false>(buf, bytes, win, 100, 16, 0, sink, ghz);
  run<128, 8, false>(buf, bytes, 200 * MB, 100, 16, 1, sink, ghz);
  // 2. by row size, L2-resident
  run<64, 8, false>(buf, bytes, 1 * MB, 100, 16, 0, sink, ghz);
  run<256, 8, false>(buf, bytes, 1 * MB, 100, 16, 0, sink, ghz);
  // 3. absent rows: do out-of-range lanes cost address cycles?
  for (int p : {61, 20, 0}) run<128, 8, false>(buf, bytes, 1 * MB, p, 16, 0, sink, ghz);
  run<64, 8, false>(buf, bytes, 1 * MB, 20, 16, 0, sink, ghz);
  // 4. by bytes in flight
  for (int w : {4, 8, 16, 32}) run<128, 4, false>(buf, bytes, 1 * MB, 100, w, 0, sink, ghz);
  run<128, 16, false>(buf, bytes, 1 * MB, 100, 16, 0, sink, ghz);
  run<128, 16, false>(buf, bytes, 2 * MB, 61, 16, 0, sink, ghz);
  // 5. LDS-DMA destination
  run<128, 8, true>(buf, bytes, 1 * MB, 100, 14, 0, sink, ghz);
  run<128, 8, true>(buf, bytes, 1 * MB, 61, 14, 0, sink, ghz);
  run<128, 8, true>(buf, bytes, 200 * MB, 100, 14, 0, sink, ghz);
  // 6. lanes per row: the direct conv kernel's A-operand pattern is ONE lane per row and 16-B piece (32 rows x 2 pieces per instruction)
  run<64, 8, false, 1>(buf, bytes, 1 * MB, 100, 16, 0, sink, ghz);
  run<64, 8, false, 2>(buf, bytes, 1 * MB, 100, 16, 0, sink, ghz);
  run<128, 8, false, 1>(buf, bytes, 1 * MB, 100, 16, 0, sink, ghz);
  run<128, 8, false, 4>(buf, bytes, 1 * MB, 100, 16, 0, sink, ghz);
  run<64, 8, false, 1>(buf, bytes, 1 * MB, 20, 16, 0, sink, ghz);
  // 7. absent lanes switched off in EXEC instead of sent out of range
  run<128, 8, false, 8, true>(buf, bytes, 1 * MB, 61, 16, 0, sink, ghz);
  run<128, 8, false, 8, true>(buf, bytes, 1 * MB, 20, 16, 0, sink, ghz);
  run<64, 8, false, 4, true>(buf, bytes, 1 * MB, 20, 16, 0, sink, ghz);
  run<64, 8, false, 1, true>(buf, bytes, 1 * MB, 20, 16, 0, sink, ghz);
  // 8. control: the same instruction streaming through an L2-resident window
  run<128, 8, false>(buf, bytes, 1 * MB, 100, 16, 1, sink, ghz);
  // 9. the MFMA-fragment shape (lane l: row l % 32, piece l / 32): what k_conv_direct / k_conv_stream issue
  run<192, 8, false, 2, false, true>(buf, bytes, 1 * MB, 88, 16, 0, sink, ghz);
  run<192, 8, false, 2, false, false>(buf, bytes, 1 * MB, 88, 16, 0, sink, ghz);
  run<192, 8, false, 4, false, false>(buf, bytes, 1 * MB, 88, 16, 0, sink, ghz);
  run<192, 8, false, 4, false, true>(buf, bytes, 1 * MB, 88, 16, 0, sink, ghz);
  run<128, 8, false, 2, false, true>(buf, bytes, 1 * MB, 61, 16, 0, sink, ghz);
  run<128, 8, false, 8, false, false>(buf, bytes, 1 * MB, 61, 16, 0, sink, ghz);
  run<64, 8, false, 2, false, true>(buf, bytes, 1 * MB, 20, 16, 0, sink, ghz);
  run<64, 8, false, 2, false, true>(buf, bytes, 1 * MB, 100, 16, 0, sink, ghz);
  return 0;
}
