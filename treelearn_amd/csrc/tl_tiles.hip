// Inference tiling on the device: one launch sequence per tile replaces the reference's
// box masks + `.cpu()` + np.savez / np.load round trip + TreeDataset.__getitem__ bookkeeping
// (tree_learn/util/data_preparation.py:393-441,456-476; tree_learn/dataset/dataset.py:34-76,87-91).
// HBM-bound: reads 4 (N + F N) bytes of the plot once per tile, writes only the kept rows.
//
// The dtype walk of the reference is kept literally (it decides which points sit on a tile edge):
//   outer square: float32 compares (torch compares a float32 tensor with 0-dim float64 tensors in float32);
//   inner-square occupancy: float64 compares (numpy float32 array vs np.float64 scalars);
//   centring: float64 subtraction of the float32-computed tile centre, stored as float32;
//   masks_inner: inf-norm of the centred float32 xy <= inner_square_edge_length / 2.
#include "tl_common.h"

namespace {

constexpr int kItems = 8, kTile = 256 * kItems;

__device__ __forceinline__ uint32_t block_scan2(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wsum[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t inc = v;
  for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, off); if (lane >= off) inc += t; }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < 4; ++w) { if (w < wid) base += wsum[w]; tot += wsum[w]; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__device__ __forceinline__ bool in_outer(const tl_tile_box& b, float x, float y) {
  return x >= b.outer[0] && x <= b.outer[1] && y >= b.outer[2] && y <= b.outer[3];
}
__device__ __forceinline__ bool in_inner(const tl_tile_box& b, float x, float y) {
  const double xd = (double)x, yd = (double)y;
  return xd >= b.inner[0] && xd < b.inner[1] && yd > b.inner[2] && yd <= b.inner[3];
}

// per block: kept rows; rows inside the inner square are summed into count[1] (at most one atomic per wave)
__global__ void __launch_bounds__(256) k_tile_partials(const float* __restrict__ xyz, int64_t n, tl_tile_box box, int32_t* __restrict__ part,
                                                       int32_t* __restrict__ count) {
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  uint32_t s = 0, in = 0;
  for (int j = 0; j < kItems; ++j)
    if (base + j < n) {
      const float x = xyz[(base + j) * 3], y = xyz[(base + j) * 3 + 1];
      if (in_outer(box, x, y)) { ++s; in += in_inner(box, x, y); }
    }
  uint32_t tot; block_scan2(s, &tot);
  if (threadIdx.x == 0) part[blockIdx.x] = (int32_t)tot;
  uint32_t tin; block_scan2(in, &tin);
  if (threadIdx.x == 0 && tin) atomicAdd(&count[1], (int32_t)tin);
}
__global__ void __launch_bounds__(256) k_tile_scan(int32_t* __restrict__ part, int64_t nb, int32_t* __restrict__ count) {
  uint32_t carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += 256) {
    const int64_t i = b0 + threadIdx.x;
    const uint32_t v = i < nb ? (uint32_t)part[i] : 0u;
    uint32_t tot; const uint32_t ex = block_scan2(v, &tot);
    if (i < nb) part[i] = (int32_t)(carry + ex);
    carry += tot;
  }
  if (threadIdx.x == 0) count[0] = (int32_t)carry;
}
__global__ void __launch_bounds__(256) k_tile_scatter(const float* __restrict__ xyz, const float* __restrict__ label, const float* __restrict__ feat,
                                                      int64_t n, int F, tl_tile_box box, const int32_t* __restrict__ part,
                                                      float* __restrict__ coords, float* __restrict__ ofeat, int64_t* __restrict__ inst,
                                                      int64_t* __restrict__ sem, uint8_t* __restrict__ m_inner, uint8_t* __restrict__ m_sem) {
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  uint32_t s = 0; bool keep[kItems];
  for (int j = 0; j < kItems; ++j) {
    keep[j] = (base + j < n) && in_outer(box, xyz[(base + j) * 3], xyz[(base + j) * 3 + 1]);
    s += keep[j];
  }
  uint32_t tot; int64_t pos = block_scan2(s, &tot) + (uint32_t)part[blockIdx.x];
  for (int j = 0; j < kItems; ++j)
    if (keep[j]) {
      const int64_t r = base + j;
      const float cx = (float)((double)xyz[r * 3] - box.center[0]), cy = (float)((double)xyz[r * 3 + 1] - box.center[1]);
      coords[pos * 3] = cx; coords[pos * 3 + 1] = cy; coords[pos * 3 + 2] = xyz[r * 3 + 2];
      for (int c = 0; c < F; ++c) ofeat[pos * F + c] = feat[r * F + c];
      const int32_t il = (int32_t)label[r];                               // .astype(np.int32), data_preparation.py:480
      inst[pos] = il;
      sem[pos] = il == 0 ? 1 : 0;                                         // dataset.py:46-48 (non-tree 0 -> class 1, everything else tree = 0)
      const bool inner = fmaxf(fabsf(cx), fabsf(cy)) <= box.half_inner;   // dataset.py:87-91
      m_inner[pos] = inner;
      m_sem[pos] = inner && il != -1;                                     // dataset.py:62,64
      ++pos;
    }
}

}  // namespace

extern "C" {

int64_t tl_tile_crop_ws_words(int64_t n) { return tl_cdiv(n, kTile) + 1; }

int tl_tile_crop(const float* xyz, const float* label, const float* feat, int64_t n, int F, const tl_tile_box* box, float* coords,
                 float* out_feat, int64_t* instance_labels, int64_t* semantic_labels, uint8_t* mask_inner, uint8_t* mask_sem,
                 int32_t* count, int32_t* ws, tl_stream_t stream) {
  if (!xyz || !label || (F > 0 && (!feat || !out_feat)) || !box || !coords || !instance_labels || !semantic_labels || !mask_inner || !mask_sem ||
      !count || !ws || n <= 0 || F < 0)
    return TL_ERR_ARG;
  const int64_t nb = tl_cdiv(n, kTile);
  hipStream_t s = tl_s(stream);
  if (hipMemsetAsync(count, 0, 2 * sizeof(int32_t), s) != hipSuccess) return TL_ERR_LAUNCH;
  k_tile_partials<<<(unsigned)nb, 256, 0, s>>>(xyz, n, *box, ws, count);
  k_tile_scan<<<1, 256, 0, s>>>(ws, nb, count);
  k_tile_scatter<<<(unsigned)nb, 256, 0, s>>>(xyz, label, feat, n, F, *box, ws, coords, out_feat, instance_labels, semantic_labels, mask_inner, mask_sem);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

}  // extern "C"
