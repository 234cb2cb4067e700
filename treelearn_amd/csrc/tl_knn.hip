// k-nearest-neighbour majority vote (assign_remaining_points_nearest_neighbor,
// reference tree_learn/util/pipeline.py:287-296 -- sklearn KNeighborsClassifier(n_neighbors=5), uniform weights).
// Exact brute force: one thread per query, reference points streamed through LDS tiles, squared distances in fp64 on
// the fp32 inputs (as sklearn's trees compute them), the k best kept sorted in registers; vote = most frequent label,
// ties -> smallest label (scipy.stats.mode).  O(nq * nr): fine up to ~1e5 x 1e6.
// tl_knn_vote_grid: the same vote, same arithmetic and same tie rules (k nearest by (distance, reference index)), over a uniform
// cell grid: the caller sorts the reference points by cell key; a query walks Chebyshev rings of cells around its own cell until its
// k-th best distance is provably smaller than anything an unvisited ring can hold.  Bit-identical to the brute-force kernel.
#include "tl_common.h"

namespace {
constexpr int kBlock = 256;
constexpr int kMaxK = 8;

template <int KK>
__global__ void __launch_bounds__(kBlock) k_knn_vote(const float* __restrict__ ref, const int64_t* __restrict__ rlab, int64_t nr,
                                                     const float* __restrict__ q, int64_t nq, int k, int64_t* __restrict__ out) {
  __shared__ float sx[kBlock], sy[kBlock], sz[kBlock];
  __shared__ int64_t sl[kBlock];
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = i < nq;
  const double x = live ? (double)q[3 * i] : 0.0, y = live ? (double)q[3 * i + 1] : 0.0, z = live ? (double)q[3 * i + 2] : 0.0;
  double bd[KK]; int64_t bl[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) { bd[t] = 1e300; bl[t] = 0; }
  for (int64_t j0 = 0; j0 < nr; j0 += kBlock) {
    const int64_t j = j0 + threadIdx.x;
    __syncthreads();
    if (j < nr) { sx[threadIdx.x] = ref[3 * j]; sy[threadIdx.x] = ref[3 * j + 1]; sz[threadIdx.x] = ref[3 * j + 2]; sl[threadIdx.x] = rlab[j]; }
    __syncthreads();
    if (!live) continue;
    const int m = (int)((nr - j0) < kBlock ? (nr - j0) : kBlock);
    for (int t = 0; t < m; ++t) {
      const double dx = __dsub_rn((double)sx[t], x), dy = __dsub_rn((double)sy[t], y), dz = __dsub_rn((double)sz[t], z);
      const double d = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
      if (d < bd[KK - 1]) {                                  // sorted insertion (strict <: earlier reference index wins ties)
        double cd = d; int64_t cl = sl[t];
#pragma unroll
        for (int u = 0; u < KK; ++u) {
          if (cd < bd[u]) { const double td = bd[u]; const int64_t tl = bl[u]; bd[u] = cd; bl[u] = cl; cd = td; cl = tl; }
        }
      }
    }
  }
  if (!live) return;
  const int kk = (int)(nr < k ? nr : k);                  // k <= KK: the vote looks at the k nearest of the KK kept
  int best_cnt = 0; int64_t best_lab = 0;
  for (int a = 0; a < kk; ++a) {
    int c = 0;
    for (int b = 0; b < kk; ++b) c += (bl[b] == bl[a]);
    if (c > best_cnt || (c == best_cnt && bl[a] < best_lab)) { best_cnt = c; best_lab = bl[a]; }
  }
  out[i] = best_lab;
}

struct KnnGrid {
  float lo[3]; float inv_h; float h;
  int dims[3];
  int64_t ncells;
};

// sorted unique cell keys -> position or -1
static __device__ __forceinline__ int64_t find_cell(const int64_t* __restrict__ keys, int64_t n, int64_t key) {
  int64_t a = 0, b = n;
  while (a < b) { const int64_t m = (a + b) >> 1; if (keys[m] < key) a = m + 1; else b = m; }
  return (a < n && keys[a] == key) ? a : -1;
}

template <int KK>
__global__ void __launch_bounds__(kBlock) k_knn_vote_grid(const float* __restrict__ ref, const int64_t* __restrict__ rlab, const int64_t* __restrict__ ridx,
                                                          const int64_t* __restrict__ ckeys, const int64_t* __restrict__ cstart, KnnGrid g,
                                                          const float* __restrict__ q, int64_t nq, int64_t nr, int k, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nq) return;
  const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
  const double x = (double)qx, y = (double)qy, z = (double)qz;
  // the query's (possibly virtual: outside the box) cell, same fp32 arithmetic as the host-side binning of the reference points
  const int cx = (int)floorf((qx - g.lo[0]) * g.inv_h), cy = (int)floorf((qy - g.lo[1]) * g.inv_h), cz = (int)floorf((qz - g.lo[2]) * g.inv_h);
  double bd[KK]; int64_t bl[KK], bi[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) { bd[t] = 1e300; bl[t] = 0; bi[t] = 0x7FFFFFFFFFFFFFFFll; }
  const int kk = (int)(nr < k ? nr : k);
  // rings beyond rmax cannot contain a cell of the grid
  const int rmax = max(max(max(cx, g.dims[0] - 1 - cx), max(cy, g.dims[1] - 1 - cy)), max(max(cz, g.dims[2] - 1 - cz), 0));
  for (int r = 0; r <= rmax; ++r) {
    for (int dx = -r; dx <= r; ++dx) {
      const int ax = cx + dx;
      if (ax < 0 || ax >= g.dims[0]) continue;
      for (int dy = -r; dy <= r; ++dy) {
        const int ay = cy + dy;
        if (ay < 0 || ay >= g.dims[1]) continue;
        const bool shell = (dx == -r || dx == r || dy == -r || dy == r);
        for (int dz = -r; dz <= r; dz += (shell ? 1 : 2 * r > 0 ? 2 * r : 1)) {      // interior (dx, dy): only the two end caps dz = +-r
          const int az = cz + dz;
          if (az < 0 || az >= g.dims[2]) continue;
          const int64_t c = find_cell(ckeys, g.ncells, ((int64_t)ax * g.dims[1] + ay) * g.dims[2] + az);
          if (c < 0) continue;
          for (int64_t j = cstart[c]; j < cstart[c + 1]; ++j) {
            const double ddx = __dsub_rn((double)ref[3 * j], x), ddy = __dsub_rn((double)ref[3 * j + 1], y), ddz = __dsub_rn((double)ref[3 * j + 2], z);
            const double d = __dadd_rn(__dadd_rn(__dmul_rn(ddx, ddx), __dmul_rn(ddy, ddy)), __dmul_rn(ddz, ddz));
            const int64_t id = ridx[j];
            if (d < bd[KK - 1] || (d == bd[KK - 1] && id < bi[KK - 1])) {       // total order (distance, original index) = the brute-force kernel's
              double cd = d; int64_t cl = rlab[j], ci = id;
#pragma unroll
              for (int u = 0; u < KK; ++u) {
                if (cd < bd[u] || (cd == bd[u] && ci < bi[u])) {
                  const double td = bd[u]; const int64_t tl = bl[u], ti = bi[u];
                  bd[u] = cd; bl[u] = cl; bi[u] = ci; cd = td; cl = tl; ci = ti;
                }
              }
            }
          }
        }
      }
    }
    // every point of a cell in ring r+1 or beyond lies more than r * h from the query (0.999: cell binning is fp32 arithmetic)
    const double reach = (double)r * (double)g.h * 0.999;
    if (bd[kk - 1] < reach * reach) break;
  }
  int best_cnt = 0; int64_t best_lab = 0;
  for (int a = 0; a < kk; ++a) {
    int c = 0;
    for (int b = 0; b < kk; ++b) c += (bl[b] == bl[a]);
    if (c > best_cnt || (c == best_cnt && bl[a] < best_lab)) { best_cnt = c; best_lab = bl[a]; }
  }
  out[i] = best_lab;
}
}  // namespace

extern "C" int tl_knn_vote(const float* ref_xyz, const int64_t* ref_label, int64_t nr, const float* q_xyz, int64_t nq, int k,
                           int64_t* out_label, tl_stream_t stream) {
  if (!ref_xyz || !ref_label || !q_xyz || !out_label || nr <= 0 || nq <= 0 || k < 1 || k > kMaxK) return TL_ERR_ARG;
  const unsigned g = (unsigned)tl_cdiv(nq, kBlock);
  hipStream_t s = tl_s(stream);
  switch (k) {
    case 1: k_knn_vote<1><<<g, kBlock, 0, s>>>(ref_xyz, ref_label, nr, q_xyz, nq, k, out_label); break;
    case 3: k_knn_vote<3><<<g, kBlock, 0, s>>>(ref_xyz, ref_label, nr, q_xyz, nq, k, out_label); break;
    case 5: k_knn_vote<5><<<g, kBlock, 0, s>>>(ref_xyz, ref_label, nr, q_xyz, nq, k, out_label); break;
    default: k_knn_vote<kMaxK><<<g, kBlock, 0, s>>>(ref_xyz, ref_label, nr, q_xyz, nq, k, out_label); break;
  }
  TL_CHECK_LAUNCH();
  return TL_OK;
}

// Reference points sorted by cell key (cell = floor((p - lo) * inv_h) per axis in fp32, key = (cx * dims[1] + cy) * dims[2] + cz),
// with their labels and ORIGINAL indices; cell_keys: the ncells distinct keys ascending, cell_start[ncells + 1] their row ranges.
extern "C" int tl_knn_vote_grid(const float* ref_sorted_xyz, const int64_t* ref_sorted_label, const int64_t* ref_sorted_index, int64_t nr,
                                const int64_t* cell_keys, const int64_t* cell_start, int64_t ncells, const float lo[3], float h, const int32_t dims[3],
                                const float* q_xyz, int64_t nq, int k, int64_t* out_label, tl_stream_t stream) {
  if (!ref_sorted_xyz || !ref_sorted_label || !ref_sorted_index || !cell_keys || !cell_start || !lo || !dims || !q_xyz || !out_label || nr <= 0 ||
      nq <= 0 || ncells <= 0 || k < 1 || k > kMaxK || !(h > 0.f))
    return TL_ERR_ARG;
  KnnGrid g;
  for (int a = 0; a < 3; ++a) { g.lo[a] = lo[a]; g.dims[a] = dims[a]; }
  g.h = h; g.inv_h = 1.0f / h; g.ncells = ncells;
  const unsigned grid = (unsigned)tl_cdiv(nq, kBlock);
  hipStream_t s = tl_s(stream);
  switch (k) {
    case 1: k_knn_vote_grid<1><<<grid, kBlock, 0, s>>>(ref_sorted_xyz, ref_sorted_label, ref_sorted_index, cell_keys, cell_start, g, q_xyz, nq, nr, k, out_label); break;
    case 3: k_knn_vote_grid<3><<<grid, kBlock, 0, s>>>(ref_sorted_xyz, ref_sorted_label, ref_sorted_index, cell_keys, cell_start, g, q_xyz, nq, nr, k, out_label); break;
    case 5: k_knn_vote_grid<5><<<grid, kBlock, 0, s>>>(ref_sorted_xyz, ref_sorted_label, ref_sorted_index, cell_keys, cell_start, g, q_xyz, nq, nr, k, out_label); break;
    default: k_knn_vote_grid<kMaxK><<<grid, kBlock, 0, s>>>(ref_sorted_xyz, ref_sorted_label, ref_sorted_index, cell_keys, cell_start, g, q_xyz, nq, nr, k, out_label); break;
  }
  TL_CHECK_LAUNCH();
  return TL_OK;
}
