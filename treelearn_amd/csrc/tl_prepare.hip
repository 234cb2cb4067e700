// Plot preparation on the device: the global voxel down-sample and the verticality feature that the
// reference delegates to open3d 0.17.0 (VoxelDownSampleAndTrace; tree_learn/util/data_preparation.py:60-79) and
// jakteristics 0.5.1 (compute_features, search radius 0.6 m; data_preparation.py:82-88).  Both libraries are absent from
// the reference tree and from this image: their published algorithms are restated (parity unpinned).
//
// Down-sample: voxel = floor((p - min_bound) / voxel_size) in double on the 2-decimal-rounded input; the host sorts the
// packed voxel keys (stable), then one thread per voxel adds its points IN INPUT ORDER in double -- the same
// left-to-right accumulation open3d's AccumulatedPoint does, so the averages are reproducible bit for bit -- and records
// the first original index (what `idx_keep` takes) and every point's voxel row (the trace, used to carry predictions back
// to the original cloud instead of the reference's Python hash dictionary, util/pipeline.py:423-452).  Voxels come out in
// ascending (x, y, z) order; open3d's order is that of an unordered_map and cannot be pinned.
//
// Verticality: points sorted by a (radius)-sized cell key with z fastest; each point scans the 9 (dx, dy) columns, whose
// three z-cells are one contiguous range of the sorted array (two binary searches), accumulates n, sum d, sum d d^T in
// double relative to itself, forms the sample covariance (n - 1), diagonalises it with cyclic Jacobi rotations and
// returns 1 - |z component of the eigenvector of the smallest eigenvalue|.  Fewer than 3 neighbours -> NaN (the caller
// replaces NaNs by the column mean like replace_nanfeatures, data_preparation.py:91-100).
#include "tl_common.h"

namespace {

constexpr int kBits = 21;
constexpr int64_t kMask = (1ll << kBits) - 1;
constexpr int kItems = 8, kTile = 256 * kItems;

__device__ __forceinline__ double round2(double v) { return rint(v * 100.0) / 100.0; }          // np.round(x, 2) on float64

__device__ __forceinline__ uint32_t block_scan3(uint32_t v, uint32_t* total) {
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

__global__ void k_cell_keys(const double* __restrict__ xyz, int64_t n, double cell, double min_bound, int64_t b0, int64_t b1, int64_t b2,
                            int round_input, int64_t* __restrict__ keys, int32_t* __restrict__ err) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t v[3];
    for (int a = 0; a < 3; ++a) {
      double p = xyz[i * 3 + a];
      if (round_input) p = round2(p);
      v[a] = (int64_t)floor((p - min_bound) / cell);
    }
    const int64_t x = v[0] - b0, y = v[1] - b1, z = v[2] - b2;
    if ((x | y | z) < 0 || x > kMask || y > kMask || z > kMask) { *err = 1; keys[i] = 0; continue; }
    keys[i] = (x << (2 * kBits)) | (y << kBits) | z;
  }
}

__global__ void __launch_bounds__(256) k_head_partials(const int64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ part) {
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  uint32_t s = 0;
  for (int j = 0; j < kItems; ++j) { const int64_t i = base + j; if (i < n) s += (i == 0 || keys[i] != keys[i - 1]); }
  uint32_t tot; block_scan3(s, &tot);
  if (threadIdx.x == 0) part[blockIdx.x] = (int32_t)tot;
}
__global__ void __launch_bounds__(256) k_head_scan(int32_t* __restrict__ part, int64_t nb, int64_t* __restrict__ count) {
  uint32_t carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += 256) {
    const int64_t i = b0 + threadIdx.x;
    const uint32_t v = i < nb ? (uint32_t)part[i] : 0u;
    uint32_t tot; const uint32_t ex = block_scan3(v, &tot);
    if (i < nb) part[i] = (int32_t)(carry + ex);
    carry += tot;
  }
  if (threadIdx.x == 0) *count = (int64_t)carry;
}
// one thread per voxel (segment of equal keys): in-order double sum, first index, trace
__global__ void __launch_bounds__(256) k_ds_reduce(const double* __restrict__ xyz, const int64_t* __restrict__ keys, const int64_t* __restrict__ perm,
                                                   int64_t n, const int32_t* __restrict__ part, float* __restrict__ out_xyz,
                                                   int64_t* __restrict__ first_idx, int64_t* __restrict__ point2vox) {
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  uint32_t s = 0; bool head[kItems];
  for (int j = 0; j < kItems; ++j) { const int64_t i = base + j; head[j] = i < n && (i == 0 || keys[i] != keys[i - 1]); s += head[j]; }
  uint32_t tot; int64_t vox = block_scan3(s, &tot) + (uint32_t)part[blockIdx.x];
  for (int j = 0; j < kItems; ++j)
    if (head[j]) {
      const int64_t i = base + j, key = keys[i];
      double sx = 0.0, sy = 0.0, sz = 0.0; int64_t cnt = 0;
      for (int64_t q = i; q < n && keys[q] == key; ++q) {
        const int64_t o = perm[q];
        sx += round2(xyz[o * 3]); sy += round2(xyz[o * 3 + 1]); sz += round2(xyz[o * 3 + 2]);
        point2vox[o] = vox; ++cnt;
      }
      const double inv = (double)cnt;
      // average in double (open3d), then .astype(float32) and np.round(., 2) in float32 (util/pipeline.py:44-45)
      const float fx = (float)(sx / inv), fy = (float)(sy / inv), fz = (float)(sz / inv);
      out_xyz[vox * 3] = rintf(fx * 100.0f) / 100.0f; out_xyz[vox * 3 + 1] = rintf(fy * 100.0f) / 100.0f; out_xyz[vox * 3 + 2] = rintf(fz * 100.0f) / 100.0f;
      first_idx[vox] = perm[i];
      ++vox;
    }
}

// one thread per group (segment of equal sorted keys): per column the double sum of the members in input order / count
__global__ void __launch_bounds__(256) k_group_mean(const float* __restrict__ src, int C, const int64_t* __restrict__ keys,
                                                    const int64_t* __restrict__ perm, int64_t n, const int32_t* __restrict__ part,
                                                    double* __restrict__ mean, int64_t* __restrict__ first_idx) {
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  uint32_t s = 0; bool head[kItems];
  for (int j = 0; j < kItems; ++j) { const int64_t i = base + j; head[j] = i < n && (i == 0 || keys[i] != keys[i - 1]); s += head[j]; }
  uint32_t tot; int64_t grp = block_scan3(s, &tot) + (uint32_t)part[blockIdx.x];
  for (int j = 0; j < kItems; ++j)
    if (head[j]) {
      const int64_t i = base + j, key = keys[i];
      int64_t cnt = 0;
      for (int64_t q = i; q < n && keys[q] == key; ++q) ++cnt;
      const double inv = 1.0 / (double)cnt;
      for (int c0 = 0; c0 < C; c0 += 8) {                      // eight columns at a time: bounded registers, rows re-read from L2
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t q = i; q < i + cnt; ++q) {
          const float* row = src + perm[q] * (int64_t)C + c0;
          for (int c = 0; c < 8; ++c) if (c0 + c < C) acc[c] += (double)row[c];
        }
        for (int c = 0; c < 8; ++c) if (c0 + c < C) mean[grp * C + c0 + c] = acc[c] * inv;
      }
      first_idx[grp] = perm[i];
      ++grp;
    }
}

__device__ __forceinline__ int64_t lower_bound(const int64_t* __restrict__ a, int64_t n, int64_t v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}

// smallest-eigenvalue eigenvector of a symmetric 3x3 matrix by cyclic Jacobi; returns its z component
__device__ double smallest_evec_z(double a00, double a01, double a02, double a11, double a12, double a22) {
  double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off <= 1e-300 || off <= 1e-18 * (fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]))) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
        for (int k = 0; k < 3; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
        for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
      }
  }
  int m = 0;
  if (A[1][1] < A[m][m]) m = 1;
  if (A[2][2] < A[m][m]) m = 2;
  return V[2][m];
}

__global__ void __launch_bounds__(256) k_verticality(const double* __restrict__ xyz, const int64_t* __restrict__ keys, int64_t n, double r2,
                                                     int64_t nx, int64_t ny, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t key = keys[i];
  const int64_t cx = key >> (2 * kBits), cy = (key >> kBits) & kMask, cz = key & kMask;
  const double px = xyz[i * 3], py = xyz[i * 3 + 1], pz = xyz[i * 3 + 2];
  double s0 = 0, s1 = 0, s2 = 0, s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0;
  int64_t cnt = 0;
  const int64_t z0 = cz > 0 ? cz - 1 : 0, z1 = cz < kMask ? cz + 1 : kMask;
  for (int64_t x = cx - 1; x <= cx + 1; ++x) {
    if (x < 0 || x > nx) continue;
    for (int64_t y = cy - 1; y <= cy + 1; ++y) {
      if (y < 0 || y > ny) continue;
      const int64_t kb = (x << (2 * kBits)) | (y << kBits);
      const int64_t lo = lower_bound(keys, n, kb | z0), hi = lower_bound(keys, n, (kb | z1) + 1);
      for (int64_t j = lo; j < hi; ++j) {
        const double dx = xyz[j * 3] - px, dy = xyz[j * 3 + 1] - py, dz = xyz[j * 3 + 2] - pz;
        if (dx * dx + dy * dy + dz * dz <= r2) {
          ++cnt; s0 += dx; s1 += dy; s2 += dz;
          s00 += dx * dx; s01 += dx * dy; s02 += dx * dz; s11 += dy * dy; s12 += dy * dz; s22 += dz * dz;
        }
      }
    }
  }
  if (cnt < 3) { out[i] = __builtin_nanf(""); return; }
  const double inv_n = 1.0 / (double)cnt, inv = 1.0 / (double)(cnt - 1);
  const double vz = smallest_evec_z((s00 - s0 * s0 * inv_n) * inv, (s01 - s0 * s1 * inv_n) * inv, (s02 - s0 * s2 * inv_n) * inv,
                                    (s11 - s1 * s1 * inv_n) * inv, (s12 - s1 * s2 * inv_n) * inv, (s22 - s2 * s2 * inv_n) * inv);
  out[i] = (float)(1.0 - fabs(vz));
}

}  // namespace

extern "C" {

int tl_cell_keys(const double* xyz, int64_t n, double cell, double min_bound, const int64_t* base3, int round_input, int64_t* keys, int32_t* err,
                 tl_stream_t stream) {
  if (!xyz || !base3 || !keys || !err || n <= 0 || !(cell > 0)) return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  if (hipMemsetAsync(err, 0, sizeof(int32_t), s) != hipSuccess) return TL_ERR_LAUNCH;
  k_cell_keys<<<tl_grid(n, 256), 256, 0, s>>>(xyz, n, cell, min_bound, base3[0], base3[1], base3[2], round_input, keys, err);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int64_t tl_downsample_ws_words(int64_t n) { return tl_cdiv(n, kTile) + 1; }

int tl_downsample_reduce(const double* xyz, const int64_t* sorted_keys, const int64_t* perm, int64_t n, float* out_xyz, int64_t* first_idx,
                         int64_t* point2vox, int64_t* n_voxels, int32_t* ws, tl_stream_t stream) {
  if (!xyz || !sorted_keys || !perm || !out_xyz || !first_idx || !point2vox || !n_voxels || !ws || n <= 0) return TL_ERR_ARG;
  const int64_t nb = tl_cdiv(n, kTile);
  hipStream_t s = tl_s(stream);
  k_head_partials<<<(unsigned)nb, 256, 0, s>>>(sorted_keys, n, ws);
  k_head_scan<<<1, 256, 0, s>>>(ws, nb, n_voxels);
  k_ds_reduce<<<(unsigned)nb, 256, 0, s>>>(xyz, sorted_keys, perm, n, ws, out_xyz, first_idx, point2vox);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_group_mean(const float* src, int64_t n, int C, const int64_t* sorted_keys, const int64_t* perm, double* mean, int64_t* first_idx,
                  int64_t* n_groups, int32_t* ws, tl_stream_t stream) {
  if (!src || !sorted_keys || !perm || !mean || !first_idx || !n_groups || !ws || n <= 0 || C <= 0) return TL_ERR_ARG;
  const int64_t nb = tl_cdiv(n, kTile);
  hipStream_t s = tl_s(stream);
  k_head_partials<<<(unsigned)nb, 256, 0, s>>>(sorted_keys, n, ws);
  k_head_scan<<<1, 256, 0, s>>>(ws, nb, n_groups);
  k_group_mean<<<(unsigned)nb, 256, 0, s>>>(src, C, sorted_keys, perm, n, ws, mean, first_idx);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_verticality(const double* xyz_sorted, const int64_t* sorted_keys, int64_t n, double radius, const int64_t* extent2, float* out,
                   tl_stream_t stream) {
  if (!xyz_sorted || !sorted_keys || !extent2 || !out || n <= 0 || !(radius > 0)) return TL_ERR_ARG;
  k_verticality<<<(unsigned)tl_cdiv(n, 256), 256, 0, tl_s(stream)>>>(xyz_sorted, sorted_keys, n, radius * radius, extent2[0], extent2[1], out);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

}  // extern "C"
