// k-nearest-neighbour majority vote (assign_remaining_points_nearest_neighbor,
// reference tree_learn/util/pipeline.py:287-296 -- sklearn KNeighborsClassifier(n_neighbors=5), uniform weights).
// Exact brute force: one thread per query, reference points streamed through LDS tiles, squared distances in fp64 on
// the fp32 inputs (as sklearn's trees compute them), the k best kept sorted in registers; vote = most frequent label,
// ties -> smallest label (scipy.stats.mode).  O(nq * nr): fine up to ~1e5 x 1e6; a cell-hash pre-filter is the next step.
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
