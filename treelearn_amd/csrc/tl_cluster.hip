// Offset-shifted grouping on the GPU: DBSCAN(eps, min_samples = 2) on 2-D points, exactly.
//
// With min_samples = 2 every point that has another point within eps is a core point, so DBSCAN's
// clusters are the connected components of the eps-graph and its noise is the isolated points
// (reference tree_learn/util/pipeline.py:173-180; sklearn numbers clusters in order of their first
// point, i.e. by the smallest point index of each component).
//
// Points are binned into eps-sized cells through an open-addressing hash of the cell key (no
// dependence on the data extent, no sort); each point scans the 3x3 cell neighbourhood and unions
// itself with every earlier point within eps (lock-free union-find, larger root hooked under the
// smaller => root = smallest index of the component).  Distances are evaluated in fp64 on the fp32
// inputs, as sklearn's KD-tree does.  HBM/latency-bound integer work; no MFMA.
#include "tl_common.h"

namespace {

constexpr int kBlock = 256;
constexpr unsigned long long kEmpty = 0xFFFFFFFFFFFFFFFFull;

struct Ws {
  unsigned long long* keys;   // [H]
  int* head;                  // [H]
  int* next;                  // [n]
  int* parent;                // [n]
  int* linked;                // [n]
  int* flag;                  // [n]  (then exclusive-scanned in place)
  int* part;                  // scan partials
  int64_t H;
  int hbits;
};

__host__ __device__ inline int64_t align16(int64_t x) { return (x + 15) & ~(int64_t)15; }

inline int64_t table_size(int64_t n) {
  int64_t h = 1024;
  while (h < 2 * n) h <<= 1;
  return h;
}

inline Ws carve(void* ws, int64_t n) {
  Ws w;
  w.H = table_size(n);
  w.hbits = 0;
  while (((int64_t)1 << w.hbits) < w.H) ++w.hbits;
  char* p = (char*)ws;
  w.keys = (unsigned long long*)p; p += align16(w.H * 8);
  w.head = (int*)p; p += align16(w.H * 4);
  w.next = (int*)p; p += align16(n * 4);
  w.parent = (int*)p; p += align16(n * 4);
  w.linked = (int*)p; p += align16(n * 4);
  w.flag = (int*)p; p += align16(n * 4);
  w.part = (int*)p;
  return w;
}

__device__ __forceinline__ unsigned long long cell_key(float x, float y, double inv_eps, int dx, int dy) {
  const int cx = (int)floor((double)x * inv_eps) + dx, cy = (int)floor((double)y * inv_eps) + dy;
  return ((unsigned long long)(unsigned)cx << 32) | (unsigned)cy;
}
__device__ __forceinline__ int64_t hash_slot(unsigned long long key, int hbits) {
  return (int64_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - hbits));
}

__global__ void __launch_bounds__(kBlock) k_init(Ws w, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < w.H; i += (int64_t)gridDim.x * blockDim.x) {
    w.keys[i] = kEmpty; w.head[i] = -1;
    if (i < n) { w.parent[i] = (int)i; w.linked[i] = 0; }
  }
}

__global__ void __launch_bounds__(kBlock) k_insert(const float* __restrict__ xy, int64_t n, double inv_eps, Ws w) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long key = cell_key(xy[2 * i], xy[2 * i + 1], inv_eps, 0, 0);
    int64_t s = hash_slot(key, w.hbits);
    while (true) {
      const unsigned long long prev = atomicCAS(&w.keys[s], kEmpty, key);
      if (prev == kEmpty || prev == key) break;
      s = (s + 1) & (w.H - 1);
    }
    w.next[i] = atomicExch(&w.head[s], (int)i);
  }
}

__device__ __forceinline__ int find_root(int* parent, int x) {
  while (true) {
    const int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == x) return x;
    x = p;
  }
}
__device__ __forceinline__ void unite(int* parent, int a, int b) {
  while (true) {
    a = find_root(parent, a); b = find_root(parent, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }          // hook the larger root under the smaller
    if (atomicCAS(&parent[a], a, b) == a) return;
  }
}

__global__ void __launch_bounds__(kBlock) k_link(const float* __restrict__ xy, int64_t n, double inv_eps, double eps2, Ws w) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float xf = xy[2 * i], yf = xy[2 * i + 1];
    const double x = xf, y = yf;
    bool any = false;
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy) {
        const unsigned long long key = cell_key(xf, yf, inv_eps, dx, dy);
        int64_t s = hash_slot(key, w.hbits);
        int j = -1;
        while (true) {
          const unsigned long long k = w.keys[s];
          if (k == key) { j = w.head[s]; break; }
          if (k == kEmpty) break;
          s = (s + 1) & (w.H - 1);
        }
        for (; j >= 0; j = w.next[j]) {
          if (j == (int)i) continue;
          const double ddx = (double)xy[2 * (int64_t)j] - x, ddy = (double)xy[2 * (int64_t)j + 1] - y;
          if (ddx * ddx + ddy * ddy <= eps2) {
            any = true;
            if (j < (int)i) unite(w.parent, (int)i, j);      // each edge once
          }
        }
      }
    if (any) w.linked[i] = 1;
  }
}

__global__ void __launch_bounds__(kBlock) k_flatten(int64_t n, Ws w) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = find_root(w.parent, (int)i);
    w.flag[i] = (r == (int)i && w.linked[i]) ? 1 : 0;
    w.linked[i] = w.linked[i] ? r : -1;                        // reuse: root of the point, or -1 for noise
  }
}

// exclusive scan of flag[] (3 passes, 2048 items per block)
constexpr int kItems = 8, kTile = kBlock * kItems;
__device__ __forceinline__ int block_scan(int v, int* total) {
  __shared__ int wsum[kBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
  for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int q = 0; q < kBlock / 64; ++q) { if (q < wid) base += wsum[q]; tot += wsum[q]; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}
__global__ void __launch_bounds__(kBlock) k_scan1(const int* __restrict__ f, int64_t n, int* __restrict__ part) {
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  int s = 0;
  for (int j = 0; j < kItems; ++j) if (base + j < n) s += f[base + j];
  int tot; block_scan(s, &tot);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(kBlock) k_scan2(int* __restrict__ part, int64_t nb, int* __restrict__ total) {
  int carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += kBlock) {
    const int64_t i = b0 + threadIdx.x;
    const int v = i < nb ? part[i] : 0;
    int tot; const int ex = block_scan(v, &tot);
    if (i < nb) part[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(kBlock) k_scan3(int* __restrict__ f, int64_t n, const int* __restrict__ part) {
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  int c[kItems]; int s = 0;
  for (int j = 0; j < kItems; ++j) { c[j] = (base + j < n) ? f[base + j] : 0; s += c[j]; }
  int tot; int ex = block_scan(s, &tot) + part[blockIdx.x];
  for (int j = 0; j < kItems; ++j) { if (base + j < n) f[base + j] = ex; ex += c[j]; }
}

__global__ void __launch_bounds__(kBlock) k_labels(int64_t n, Ws w, int32_t* __restrict__ labels) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = w.linked[i];
    labels[i] = r >= 0 ? w.flag[r] : -1;                      // flag[] now holds the exclusive scan = cluster id of root r
  }
}

}  // namespace

extern "C" {

int64_t tl_cluster_ws_bytes(int64_t n) {
  if (n <= 0) return 0;
  const int64_t H = table_size(n);
  return align16(H * 8) + align16(H * 4) + 4 * align16(n * 4) + align16((tl_cdiv(n, kTile) + 1) * 4) + 64;
}

int tl_cluster_grid(const float* xy, int64_t n, double eps, int32_t* labels, int32_t* n_clusters, void* ws, tl_stream_t stream) {
  if (!xy || !labels || !n_clusters || !ws || n <= 0 || n > 0x7FFFFFF0 || !(eps > 0.0)) return TL_ERR_ARG;
  const Ws w = carve(ws, n);
  hipStream_t s = tl_s(stream);
  const double inv = 1.0 / (eps * (1.0 + 1e-9));          // cells a hair larger than eps: points within eps are always in adjacent cells
  k_init<<<tl_grid(w.H, kBlock), kBlock, 0, s>>>(w, n);
  k_insert<<<tl_grid(n, kBlock), kBlock, 0, s>>>(xy, n, inv, w);
  k_link<<<(unsigned)tl_cdiv(n, kBlock), kBlock, 0, s>>>(xy, n, inv, eps * eps, w);
  k_flatten<<<tl_grid(n, kBlock), kBlock, 0, s>>>(n, w);
  const int64_t nb = tl_cdiv(n, kTile);
  k_scan1<<<(unsigned)nb, kBlock, 0, s>>>(w.flag, n, w.part);
  k_scan2<<<1, kBlock, 0, s>>>(w.part, nb, n_clusters);
  k_scan3<<<(unsigned)nb, kBlock, 0, s>>>(w.flag, n, w.part);
  k_labels<<<tl_grid(n, kBlock), kBlock, 0, s>>>(n, w, labels);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

}  // extern "C"
