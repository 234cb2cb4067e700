// Developer micro-benchmarks (not part of the product path): how fast can a workgroup-tiled kernel pull the
// rulebook's gathered rows through L2/MALL at all?  Upper bound for any output-stationary conv without reuse.
#include "tl_common.h"

namespace {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// mode 0: every thread issues its 16-B loads for `TAPS` taps back to back (max memory-level parallelism)
template <int VPR, int TAPS>
__global__ void __launch_bounds__(256) k_gather_bench(const char* __restrict__ in, int ld_b, const int32_t* __restrict__ table, int K,
                                                      int64_t n, uint32_t* __restrict__ sink) {
  constexpr int RPP = 256 / VPR, APASS = 128 / RPP;
  const int tid = threadIdx.x, lrow = tid / VPR, cv = tid % VPR;
  const int64_t r0 = (int64_t)blockIdx.x * 128;
  const int64_t in_bytes = (n - 1) * (int64_t)ld_b + VPR * 16;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in), 0, (int)in_bytes, 0x00020000);
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int k0 = 0; k0 < K; k0 += TAPS) {
    int idx[TAPS][APASS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        const int64_t row = r0 + lrow + RPP * i;
        idx[t][i] = (k0 + t < K && row < n) ? table[(int64_t)(k0 + t) * n + row] : -1;
      }
    u32x4 v[TAPS][APASS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int i = 0; i < APASS; ++i)
        v[t][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((unsigned)idx[t][i] * (unsigned)ld_b + cv * 16), 0, 0));
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int i = 0; i < APASS; ++i) acc ^= v[t][i];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;     // keep the loads alive
}
// fragment-shaped gathers: a wave's lane (i = lane & 31, h = lane >> 5) reads 16 B at row i, byte (32*j + 16*h): exactly what a
// 32x32x16 MFMA A-operand needs -- 32 rows x 32 B per instruction (vs 8 rows x 128 B for the staging pattern)
template <int CHUNKS, int TAPS>
__global__ void __launch_bounds__(256) k_gather_frag(const char* __restrict__ in, int ld_b, const int32_t* __restrict__ table, int K,
                                                     int64_t n, uint32_t* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 128 + wv * 32 + (lane & 31);
  const int64_t in_bytes = (n - 1) * (int64_t)ld_b + CHUNKS * 32;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in), 0, (int)in_bytes, 0x00020000);
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int k0 = 0; k0 < K; k0 += TAPS) {
    int idx[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) idx[t] = (k0 + t < K && row < n) ? table[(int64_t)(k0 + t) * n + row] : -1;
    u32x4 v[TAPS][CHUNKS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int j = 0; j < CHUNKS; ++j)
        v[t][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((unsigned)idx[t] * (unsigned)ld_b + j * 32 + (lane >> 5) * 16), 0, 0));
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int j = 0; j < CHUNKS; ++j) acc ^= v[t][j];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}
// quad pattern: four adjacent lanes read 64 contiguous bytes of one row; lane (q = (lane >> 2) & 7, h = lane >> 5) reads row q + 8 i,
// bytes 64 h + 16 (lane & 3): 8 rows x 128 B per instruction like the staging pattern, but only quads are contiguous
template <int TAPS>
__global__ void __launch_bounds__(256) k_gather_quad(const char* __restrict__ in, int ld_b, const int32_t* __restrict__ table, int K,
                                                     int64_t n, uint32_t* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wv * 32 + ((lane >> 2) & 7);
  const int64_t in_bytes = (n - 1) * (int64_t)ld_b + 128;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in), 0, (int)in_bytes, 0x00020000);
  const int off = (lane >> 5) * 64 + (lane & 3) * 16;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int k0 = 0; k0 < K; k0 += TAPS) {
    int idx[TAPS][4];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) idx[t][i] = (k0 + t < K && row0 + 8 * i < n) ? table[(int64_t)(k0 + t) * n + row0 + 8 * i] : -1;
    u32x4 v[TAPS][4];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[t][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((unsigned)idx[t][i] * (unsigned)ld_b + off), 0, 0));
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc ^= v[t][i];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}
}  // namespace

extern "C" int tl_dev_gather_quad(const void* in, int64_t ld_bytes, int row_bytes, const int32_t* table, int K, int64_t n, int taps_in_flight,
                                  uint32_t* sink, tl_stream_t stream) {
  if (row_bytes != 128) return TL_ERR_ARG;
  const unsigned g = (unsigned)tl_cdiv(n, 128);
  hipStream_t s = tl_s(stream);
  if (taps_in_flight >= 9) k_gather_quad<9><<<g, 256, 0, s>>>((const char*)in, (int)ld_bytes, table, K, n, sink);
  else if (taps_in_flight >= 3) k_gather_quad<3><<<g, 256, 0, s>>>((const char*)in, (int)ld_bytes, table, K, n, sink);
  else k_gather_quad<1><<<g, 256, 0, s>>>((const char*)in, (int)ld_bytes, table, K, n, sink);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

extern "C" int tl_dev_gather_frag(const void* in, int64_t ld_bytes, int row_bytes, const int32_t* table, int K, int64_t n, int taps_in_flight,
                                  uint32_t* sink, tl_stream_t stream) {
  const unsigned g = (unsigned)tl_cdiv(n, 128);
  hipStream_t s = tl_s(stream);
#define GF(C_, T_) k_gather_frag<C_, T_><<<g, 256, 0, s>>>((const char*)in, (int)ld_bytes, table, K, n, sink)
  if (row_bytes == 64) { if (taps_in_flight >= 9) GF(2, 9); else if (taps_in_flight >= 3) GF(2, 3); else GF(2, 1); }
  else if (row_bytes == 128) { if (taps_in_flight >= 9) GF(4, 9); else if (taps_in_flight >= 3) GF(4, 3); else GF(4, 1); }
  else if (row_bytes == 192) { if (taps_in_flight >= 3) GF(6, 3); else GF(6, 1); }
  else return TL_ERR_ARG;
#undef GF
  TL_CHECK_LAUNCH();
  return TL_OK;
}

extern "C" int tl_dev_gather_bench(const void* in, int64_t ld_bytes, int row_bytes, const int32_t* table, int K, int64_t n, int taps_in_flight,
                                   uint32_t* sink, tl_stream_t stream) {
  const unsigned g = (unsigned)tl_cdiv(n, 128);
  hipStream_t s = tl_s(stream);
#define GB(VPR_, T_) k_gather_bench<VPR_, T_><<<g, 256, 0, s>>>((const char*)in, (int)ld_bytes, table, K, n, sink)
  if (row_bytes == 64) { if (taps_in_flight >= 9) GB(4, 9); else if (taps_in_flight >= 3) GB(4, 3); else GB(4, 1); }
  else if (row_bytes == 128) { if (taps_in_flight >= 9) GB(8, 9); else if (taps_in_flight >= 3) GB(8, 3); else GB(8, 1); }
  else if (row_bytes == 192) { if (taps_in_flight >= 3) GB(12, 3); else GB(12, 1); }
  else return TL_ERR_ARG;
#undef GB
  TL_CHECK_LAUNCH();
  return TL_OK;
}
