// Shared helpers for libtreelearn_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/treelearn_hip.h"

#define TL_CHECK_LAUNCH()                                   \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return TL_ERR_LAUNCH; \
  } while (0)

static inline hipStream_t tl_s(tl_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE attribute of a kernel: remembered per kernel instantiation (one static
// TlAttrOnce in its launcher) and per device ordinal, so that a process that drives several GPUs sets it on each of them.
struct TlAttrOnce {
  std::atomic<uint64_t> done[4];                  // device ordinals 0..255
};
static inline bool tl_lds_attr(TlAttrOnce& once, const void* kernel, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::atomic<uint64_t>& w = once.done[(dev >> 6) & 3];
  const uint64_t bit = 1ull << (dev & 63);
  if (w.load(std::memory_order_acquire) & bit) return true;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  w.fetch_or(bit, std::memory_order_release);
  return true;
}

static inline int64_t tl_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid for a grid-stride memory-bound kernel: enough blocks to fill 256 CUs x 8, never more than the work.
static inline unsigned tl_grid(int64_t work_items, int block) {
  int64_t g = tl_cdiv(work_items, block);
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (unsigned)g;
}

struct TlDims {  // {B, X, Y, Z} of a bitmap grid; Zw = words per z column
  int B, X, Y, Z, Zw;
};
static inline TlDims tl_dims(const int32_t d[4]) {
  TlDims r{d[0], d[1], d[2], d[3], (d[3] + 63) >> 6};
  return r;
}
__host__ __device__ static inline int64_t tl_nwords(const TlDims& d) { return (int64_t)d.B * d.X * d.Y * d.Zw; }

__device__ __forceinline__ int64_t tl_col_word(const TlDims& d, int b, int x, int y) {
  return (((int64_t)b * d.X + x) * d.Y + y) * d.Zw;
}

// rank of cell (column word base wc, z) or -1 if the bit is clear
__device__ __forceinline__ int tl_rank_at(const uint64_t* __restrict__ bm, const uint32_t* __restrict__ pf,
                                          int64_t wc, int z) {
  const int64_t w = wc + (z >> 6);
  const uint64_t word = bm[w];
  const uint64_t bit = 1ull << (z & 63);
  if (!(word & bit)) return -1;
  return (int)(pf[w] + __popcll(word & (bit - 1)));
}

bool tl_conv_one_hot_direct_enabled(int64_t n_out);     // tl_conv.hip: the gather-once (one-hot) form of the direct kernel is switched on (tl_set_tuning)
