// HDBSCAN device stage that scales (reference tree_learn/util/pipeline.py:184-191; sklearn HDBSCAN(min_cluster_size = m)):
// the same two quantities as tl_hdbscan.hip -- k-th-neighbour core distances and a minimum spanning tree of the complete
// mutual-reachability graph, fp64 on the fp32 points, the same rounding (no fused multiply-add) -- computed through a
// pointerless quadtree over a uniform cell grid instead of two O(n^2) passes:
//   * points are counting-sorted by leaf cell; every tree node carries "empty / all points in ONE component / mixed" and the
//     smallest core distance below it;
//   * core distances: nearest-first descent per point, pruned by the k-th best distance so far (exact: the k-th smallest value
//     does not depend on the visiting order);
//   * MST: Boruvka.  Per round every point looks for its lightest edge into another component (descent pruned by its own best,
//     by its component's best so far, by single-component nodes and by the node's smallest core distance), the component takes the
//     minimum under the strict total order (weight, smaller original index, larger original index), and the picks are merged with a
//     lock-free union-find.  A strict total order makes the picks cycle-free, so the result is THE minimum spanning tree of that
//     order: deterministic, and its weight multiset equals that of sklearn's Prim tree (every MST has the same weights); only the
//     choice among equal-weight edges differs from Prim's insertion order.
// The Prim form (tl_hdbscan_mst) stays: it reproduces sklearn's edge order exactly and is the test oracle of this file.
#include "tl_common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace {

constexpr int kBlock = 256;
constexpr int kMaxK = 128;
constexpr int kMaxLevel = 13;                    // leaf grid at most 8192 x 8192
constexpr int kStack = 4 * kMaxLevel + 4;
constexpr unsigned long long kNone = 0xFFFFFFFFFFFFFFFFull;
constexpr int kEmptyTag = -2, kMixedTag = -1;

struct Grid { double lox, loy, h, slack; int L; };   // leaf cell edge h, G = 1 << L cells per axis; slack = absolute round-off allowance of the binning

__host__ __device__ inline int64_t lvl_off(int l) { return ((((int64_t)1) << (2 * l)) - 1) / 3; }
__host__ __device__ inline int64_t a16(int64_t x) { return (x + 15) & ~(int64_t)15; }

__device__ __forceinline__ double dist2(double ax, double ay, double bx, double by) {
  const double dx = __dsub_rn(ax, bx), dy = __dsub_rn(ay, by);
  return __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
}
__device__ __forceinline__ unsigned fenc(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
inline float fdec(unsigned u) { u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; std::memcpy(&f, &u, 4); return f; }

// ---------------------------------------------------------------- planning: bounding box + occupancy at a probe resolution
__global__ void __launch_bounds__(kBlock) k_bbox(const float* __restrict__ xy, int64_t n, unsigned* __restrict__ mm /*minx miny maxx maxy*/) {
  unsigned a = 0xFFFFFFFFu, b = 0xFFFFFFFFu, c = 0, d = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const unsigned ex = fenc(xy[2 * i]), ey = fenc(xy[2 * i + 1]);
    a = min(a, ex); b = min(b, ey); c = max(c, ex); d = max(d, ey);
  }
  for (int off = 32; off > 0; off >>= 1) {
    a = min(a, (unsigned)__shfl_xor((int)a, off)); b = min(b, (unsigned)__shfl_xor((int)b, off));
    c = max(c, (unsigned)__shfl_xor((int)c, off)); d = max(d, (unsigned)__shfl_xor((int)d, off));
  }
  if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], a); atomicMin(&mm[1], b); atomicMax(&mm[2], c); atomicMax(&mm[3], d); }
}

__device__ __forceinline__ void cell_of(const Grid& g, double x, double y, int& cx, int& cy) {
  const int G = 1 << g.L;
  cx = (int)floor((x - g.lox) / g.h); cy = (int)floor((y - g.loy) / g.h);
  cx = cx < 0 ? 0 : (cx >= G ? G - 1 : cx); cy = cy < 0 ? 0 : (cy >= G ? G - 1 : cy);
}

__global__ void __launch_bounds__(kBlock) k_probe(const float* __restrict__ xy, int64_t n, Grid g, unsigned* __restrict__ bits) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    int cx, cy; cell_of(g, (double)xy[2 * i], (double)xy[2 * i + 1], cx, cy);
    const int64_t c = ((int64_t)cy << g.L) + cx;
    atomicOr(&bits[c >> 5], 1u << (c & 31));
  }
}
__global__ void __launch_bounds__(kBlock) k_popcount(const unsigned* __restrict__ bits, int64_t nw, unsigned* __restrict__ total) {
  unsigned s = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nw; i += (int64_t)gridDim.x * kBlock) s += __popc(bits[i]);
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor((int)s, off);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(total, s);
}

// ---------------------------------------------------------------- counting sort by leaf cell
__global__ void __launch_bounds__(kBlock) k_count(const float* __restrict__ xy, int n, Grid g, int* __restrict__ cell, int* __restrict__ cnt) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    int cx, cy; cell_of(g, (double)xy[2 * (int64_t)i], (double)xy[2 * (int64_t)i + 1], cx, cy);
    const int c = (cy << g.L) + cx;
    cell[i] = c; atomicAdd(&cnt[c], 1);
  }
}
__global__ void __launch_bounds__(kBlock) k_scatter(const float* __restrict__ xy, int n, const int* __restrict__ cell, const int* __restrict__ start,
                                                    int* __restrict__ cursor, double* __restrict__ sx, double* __restrict__ sy, int* __restrict__ oid) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const int c = cell[i];
    const int p = start[c] + atomicAdd(&cursor[c], 1);
    sx[p] = (double)xy[2 * (int64_t)i]; sy[p] = (double)xy[2 * (int64_t)i + 1]; oid[p] = i;
  }
}

// exclusive scan of int[n] -> out[n + 1] (tile = 2048 items)
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
__global__ void __launch_bounds__(kBlock) k_scan2(int* __restrict__ part, int64_t nb) {
  int carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += kBlock) {
    const int64_t i = b0 + threadIdx.x;
    const int v = i < nb ? part[i] : 0;
    int tot; const int ex = block_scan(v, &tot);
    if (i < nb) part[i] = carry + ex;
    carry += tot;
  }
}
__global__ void __launch_bounds__(kBlock) k_scan3(const int* __restrict__ f, int64_t n, const int* __restrict__ part, int* __restrict__ out) {
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  int c[kItems]; int s = 0;
  for (int j = 0; j < kItems; ++j) { c[j] = (base + j < n) ? f[base + j] : 0; s += c[j]; }
  int tot; int ex = block_scan(s, &tot) + part[blockIdx.x];
  for (int j = 0; j < kItems; ++j) { if (base + j < n) out[base + j] = ex; ex += c[j]; }
  if (base <= n - 1 && n - 1 < base + kItems) out[n] = ex;        // the thread holding the last item also writes the total
}

// ---------------------------------------------------------------- node annotations
// leaf tags from the component labels: empty / the one component of the cell / mixed
__global__ void __launch_bounds__(kBlock) k_tags_leaf(const int* __restrict__ start, int64_t ncell, const int* __restrict__ comp, int* __restrict__ tag) {
  for (int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x; c < ncell; c += (int64_t)gridDim.x * kBlock) {
    const int a = start[c], b = start[c + 1];
    int t = kEmptyTag;
    if (b > a) {
      t = comp ? comp[a] : kMixedTag;
      for (int p = a + 1; p < b && t >= 0; ++p) if (comp[p] != t) t = kMixedTag;
    }
    tag[c] = t;
  }
}
__global__ void __launch_bounds__(kBlock) k_tags_up(const int* __restrict__ child, int* __restrict__ tag, int l /* level being written */) {
  const int64_t nn = (int64_t)1 << (2 * l);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nn; i += (int64_t)gridDim.x * kBlock) {
    const int x = (int)(i & ((1 << l) - 1)), y = (int)(i >> l);
    int t = kEmptyTag;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ct = child[(((int64_t)(2 * y + (q >> 1))) << (l + 1)) + 2 * x + (q & 1)];
      if (ct == kEmptyTag) continue;
      t = (t == kEmptyTag) ? ct : (t == ct ? t : kMixedTag);
    }
    tag[i] = t;
  }
}
__global__ void __launch_bounds__(kBlock) k_mincore_leaf(const int* __restrict__ start, int64_t ncell, const double* __restrict__ core, float* __restrict__ mc) {
  for (int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x; c < ncell; c += (int64_t)gridDim.x * kBlock) {
    double m = std::numeric_limits<double>::infinity();
    for (int p = start[c]; p < start[c + 1]; ++p) m = fmin(m, core[p]);
    mc[c] = __double2float_rd(m);                              // rounded down: stays a lower bound
  }
}
__global__ void __launch_bounds__(kBlock) k_mincore_up(const float* __restrict__ child, float* __restrict__ mc, int l) {
  const int64_t nn = (int64_t)1 << (2 * l);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nn; i += (int64_t)gridDim.x * kBlock) {
    const int x = (int)(i & ((1 << l) - 1)), y = (int)(i >> l);
    float m = std::numeric_limits<float>::infinity();
#pragma unroll
    for (int q = 0; q < 4; ++q) m = fminf(m, child[(((int64_t)(2 * y + (q >> 1))) << (l + 1)) + 2 * x + (q & 1)]);
    mc[i] = m;
  }
}

// squared distance from (qx, qy) to node (l, x, y), shrunk (absolutely and relatively) so that cell-binning round-off can never make it too large
__device__ __forceinline__ double node_d2(const Grid& g, int l, int x, int y, double qx, double qy) {
  const double s = g.h * (double)(1 << (g.L - l));
  const double x0 = g.lox + s * x, y0 = g.loy + s * y;
  const double dx = fmax(fmax(x0 - qx, qx - (x0 + s)) - g.slack, 0.0), dy = fmax(fmax(y0 - qy, qy - (y0 + s)) - g.slack, 0.0);
  return (dx * dx + dy * dy) * (1.0 - 1e-8);
}
// children of (l, x, y), farthest first (so the nearest is popped first)
__device__ __forceinline__ void push_children(const Grid& g, int l, int x, int y, double qx, double qy, int* stack, int& sp) {
  const double s = g.h * (double)(1 << (g.L - l)), mx = g.lox + s * (x + 0.5), my = g.loy + s * (y + 0.5);
  const int a = qx >= mx, b = qy >= my;
  const bool x_first = fabs(qx - mx) < fabs(qy - my);          // crossing the nearer centre line comes first
  const int l1 = l + 1;
  auto code = [&](int ca, int cb) { return (l1 << 26) | ((2 * y + cb) << 13) | (2 * x + ca); };
  stack[sp++] = code(1 - a, 1 - b);
  stack[sp++] = x_first ? code(a, 1 - b) : code(1 - a, b);
  stack[sp++] = x_first ? code(1 - a, b) : code(a, 1 - b);
  stack[sp++] = code(a, b);
}

// ---------------------------------------------------------------- core distances
__global__ void __launch_bounds__(kBlock) k_core(const double* __restrict__ sx, const double* __restrict__ sy, int n, int k, Grid g,
                                                 const int* __restrict__ start, const int* __restrict__ tag, double* __restrict__ core) {
  const int p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const double x = sx[p], y = sy[p];
  double best[kMaxK];
  int cnt = 0; double worst = -1.0; int worst_at = 0;
  int stack[kStack]; int sp = 0;
  stack[sp++] = 0;
  while (sp > 0) {
    const int c = stack[--sp];
    const int l = c >> 26, nx = c & 8191, ny = (c >> 13) & 8191;
    const int64_t ni = lvl_off(l) + ((int64_t)ny << l) + nx;
    if (tag[ni] == kEmptyTag) continue;
    if (cnt >= k && node_d2(g, l, nx, ny, x, y) > worst) continue;
    if (l < g.L) { push_children(g, l, nx, ny, x, y, stack, sp); continue; }
    const int64_t cell = ((int64_t)ny << l) + nx;
    for (int j = start[cell]; j < start[cell + 1]; ++j) {
      const double d = dist2(x, y, sx[j], sy[j]);
      if (cnt < k) {
        best[cnt] = d;
        if (d > worst) { worst = d; worst_at = cnt; }
        ++cnt;
      } else if (d < worst) {
        best[worst_at] = d;
        worst = -1.0;
        for (int q = 0; q < k; ++q) if (best[q] > worst) { worst = best[q]; worst_at = q; }
      }
    }
  }
  core[p] = sqrt(cnt >= k ? worst : std::numeric_limits<double>::infinity());
}

// The same for k > kMaxK (reference util/pipeline.py:184-191 hands any tau_min to HDBSCAN(min_cluster_size)): the k smallest squared
// distances of a point live in a binary MAX-heap in the workspace (heap[q * T + t] for thread t of T = kBigThreads: coalesced), so a
// replacement costs log k instead of a rescan; a thread walks points t, t + T, ...  The value returned (the k-th smallest distance) is
// that of k_core bit for bit: only the bookkeeping differs.
constexpr int kBigThreads = 256 * 64;
constexpr int kBigMaxK = 4096;
__global__ void __launch_bounds__(kBlock) k_core_big(const double* __restrict__ sx, const double* __restrict__ sy, int n, int k, Grid g,
                                                     const int* __restrict__ start, const int* __restrict__ tag, double* __restrict__ core,
                                                     double* __restrict__ lists) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= kBigThreads) return;
  double* heap = lists + t;
  auto H = [&](int q) -> double& { return heap[(int64_t)q * kBigThreads]; };
  auto sift = [&](int i, int cnt) {
    const double v = H(i);
    while (true) {
      int c = 2 * i + 1;
      if (c >= cnt) break;
      if (c + 1 < cnt && H(c + 1) > H(c)) ++c;
      if (!(H(c) > v)) break;
      H(i) = H(c); i = c;
    }
    H(i) = v;
  };
  for (int p = t; p < n; p += kBigThreads) {
    const double x = sx[p], y = sy[p];
    int cnt = 0;
    int stack[kStack]; int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
      const int c = stack[--sp];
      const int l = c >> 26, nx = c & 8191, ny = (c >> 13) & 8191;
      const int64_t ni = lvl_off(l) + ((int64_t)ny << l) + nx;
      if (tag[ni] == kEmptyTag) continue;
      if (cnt >= k && node_d2(g, l, nx, ny, x, y) > H(0)) continue;
      if (l < g.L) { push_children(g, l, nx, ny, x, y, stack, sp); continue; }
      const int64_t cell = ((int64_t)ny << l) + nx;
      for (int j = start[cell]; j < start[cell + 1]; ++j) {
        const double d = dist2(x, y, sx[j], sy[j]);
        if (cnt < k) {
          H(cnt) = d; ++cnt;
          if (cnt == k) for (int i = k / 2 - 1; i >= 0; --i) sift(i, k);      // heapify once the list is full
        } else if (d < H(0)) {
          H(0) = d; sift(0, k);
        }
      }
    }
    core[p] = sqrt(cnt >= k ? H(0) : std::numeric_limits<double>::infinity());
  }
}

// ---------------------------------------------------------------- Boruvka
__device__ __forceinline__ int find_root(int* parent, int x) {
  while (true) {
    const int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == x) return x;
    x = p;
  }
}
__device__ __forceinline__ bool unite(int* parent, int a, int b) {
  while (true) {
    a = find_root(parent, a); b = find_root(parent, b);
    if (a == b) return false;
    if (a < b) { const int t = a; a = b; b = t; }
    if (atomicCAS(&parent[a], a, b) == a) return true;
  }
}
__device__ __forceinline__ unsigned long long edge_key(int oa, int ob) {
  const unsigned lo = (unsigned)min(oa, ob), hi = (unsigned)max(oa, ob);
  return ((unsigned long long)lo << 32) | hi;
}

// lightest edge from point p into another component: (weight, key) minimal; weight = max(core_p, core_j, |p - j|)
__global__ void __launch_bounds__(kBlock) k_search(const double* __restrict__ sx, const double* __restrict__ sy, const double* __restrict__ core,
                                                   const int* __restrict__ oid, const int* __restrict__ comp, int n, Grid g,
                                                   const int* __restrict__ start, const int* __restrict__ tag, const float* __restrict__ mincore,
                                                   unsigned long long* __restrict__ comp_best, double* __restrict__ pv, int* __restrict__ pj) {
  const int p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const double x = sx[p], y = sy[p], cp = core[p];
  const int me = comp[p], op = oid[p];
  double bv = std::numeric_limits<double>::infinity(); unsigned long long bk = kNone; int bj = -1;
  double cb = __longlong_as_double((long long)__hip_atomic_load(&comp_best[me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));   // +inf until some point of the component reports (k_flatten)
  int stack[kStack]; int sp = 0, pops = 0;
  stack[sp++] = 0;
  while (sp > 0) {
    if ((++pops & 7) == 0) cb = __longlong_as_double((long long)__hip_atomic_load(&comp_best[me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const double bound = fmin(bv, cb);
    if (cp > bound) break;                                        // every edge of p weighs at least core_p
    const int c = stack[--sp];
    const int l = c >> 26, nx = c & 8191, ny = (c >> 13) & 8191;
    const int64_t ni = lvl_off(l) + ((int64_t)ny << l) + nx;
    const int t = tag[ni];
    if (t == kEmptyTag || t == me) continue;
    if ((double)mincore[ni] > bound) continue;
    if (node_d2(g, l, nx, ny, x, y) > bound * bound) continue;    // strict: nodes that can only tie are still visited (tie-break by key)
    if (l < g.L) { push_children(g, l, nx, ny, x, y, stack, sp); continue; }
    const int64_t cell = ((int64_t)ny << l) + nx;
    for (int j = start[cell]; j < start[cell + 1]; ++j) {
      if (comp[j] == me) continue;
      const double v = fmax(fmax(cp, core[j]), sqrt(dist2(x, y, sx[j], sy[j])));
      if (v > bv) continue;
      const unsigned long long key = edge_key(op, oid[j]);
      if (v < bv || key < bk) {
        if (v < bv) atomicMin(&comp_best[me], (unsigned long long)__double_as_longlong(v));   // non-negative doubles order like their bit patterns
        bv = v; bk = key; bj = j;
      }
    }
  }
  pv[p] = bv; pj[p] = bj;
}
__global__ void __launch_bounds__(kBlock) k_win(const double* __restrict__ pv, const int* __restrict__ pj, const int* __restrict__ oid, const int* __restrict__ comp, int n,
                                                const unsigned long long* __restrict__ comp_best, unsigned long long* __restrict__ win) {
  for (int p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) {
    const int j = pj[p];
    if (j < 0) continue;
    const int me = comp[p];
    if ((unsigned long long)__double_as_longlong(pv[p]) == comp_best[me]) atomicMin(&win[me], edge_key(oid[p], oid[j]));
  }
}
__global__ void __launch_bounds__(kBlock) k_merge(const double* __restrict__ pv, const int* __restrict__ pj, const int* __restrict__ oid, const int* __restrict__ comp, int n,
                                                  const unsigned long long* __restrict__ comp_best, const unsigned long long* __restrict__ win, int* __restrict__ parent,
                                                  int* __restrict__ n_edges, int32_t* __restrict__ e_src, int32_t* __restrict__ e_dst, double* __restrict__ e_w) {
  for (int p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) {
    const int j = pj[p];
    if (j < 0) continue;
    const int me = comp[p];
    if ((unsigned long long)__double_as_longlong(pv[p]) != comp_best[me] || edge_key(oid[p], oid[j]) != win[me]) continue;
    if (unite(parent, p, j)) {                                   // fails only when the other component picked the same edge first
      const int e = atomicAdd(n_edges, 1);
      e_src[e] = min(oid[p], oid[j]); e_dst[e] = max(oid[p], oid[j]); e_w[e] = pv[p];
    }
  }
}
__global__ void __launch_bounds__(kBlock) k_flatten(int* __restrict__ parent, int* __restrict__ comp, int n, unsigned long long* __restrict__ comp_best,
                                                    unsigned long long* __restrict__ win) {
  for (int p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) {
    comp[p] = find_root(parent, p);
    comp_best[p] = 0x7FF0000000000000ull;                        // +inf
    win[p] = kNone;
  }
}
__global__ void __launch_bounds__(kBlock) k_init_uf(int* __restrict__ parent, int n) {
  for (int p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) parent[p] = p;
}
__global__ void __launch_bounds__(kBlock) k_unsort(const double* __restrict__ core_sorted, const int* __restrict__ oid, int n, double* __restrict__ core_out) {
  for (int p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) core_out[oid[p]] = core_sorted[p];
}

struct Ws {
  int *cell, *cnt, *start, *part, *tag, *oid, *comp, *parent, *pj, *n_edges;
  float* mincore;
  double *sx, *sy, *core, *pv;
  unsigned long long *comp_best, *win;
  int64_t bytes;
};
inline Ws carve(void* ws, int64_t n, int L) {
  const int64_t ncell = (int64_t)1 << (2 * L), nnode = lvl_off(L + 1);
  char* p = (char*)ws;
  Ws w;
  auto take = [&](int64_t b) { char* r = p; p += a16(b); return r; };
  w.cell = (int*)take(n * 4); w.cnt = (int*)take(ncell * 4); w.start = (int*)take((ncell + 1) * 4); w.part = (int*)take((tl_cdiv(ncell, kTile) + 1) * 4);
  w.tag = (int*)take(nnode * 4); w.mincore = (float*)take(nnode * 4);
  w.oid = (int*)take(n * 4); w.comp = (int*)take(n * 4); w.parent = (int*)take(n * 4); w.pj = (int*)take(n * 4); w.n_edges = (int*)take(16);
  w.sx = (double*)take(n * 8); w.sy = (double*)take(n * 8); w.core = (double*)take(n * 8); w.pv = (double*)take(n * 8);
  w.comp_best = (unsigned long long*)take(n * 8); w.win = (unsigned long long*)take(n * 8);
  w.bytes = p - (char*)ws + 64;
  return w;
}
inline bool grid_ok(const TlHdbGrid* g) { return g && g->levels >= 0 && g->levels <= kMaxLevel && g->h > 0.0 && std::isfinite(g->h) && std::isfinite(g->lo[0]) && std::isfinite(g->lo[1]); }

}  // namespace

extern "C" {

int64_t tl_hdbscan_grid_plan_ws_bytes(void) { return 64 + (((int64_t)1 << 20) / 8); }   // bounding box + a 1024 x 1024 occupancy bitmap

// Chooses the leaf grid for xy f32[n,2] (device): bounding box, then the leaf level such that an occupied cell holds about eight
// points (measured at a probe resolution).  SYNCHRONISES the stream twice (it reads the box and the probe count back).
int tl_hdbscan_grid_plan(const float* xy, int64_t n, TlHdbGrid* out, void* ws, tl_stream_t stream) {
  if (!xy || !out || !ws || n < 2 || n > 0x7FFFFFF0) return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  unsigned* mm = (unsigned*)ws;
  unsigned* bits = (unsigned*)((char*)ws + 64);
  const unsigned init[5] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  if (hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, s) != hipSuccess) return TL_ERR_LAUNCH;
  k_bbox<<<tl_grid(n, kBlock), kBlock, 0, s>>>(xy, n, mm);
  unsigned h4[5];
  if (hipMemcpyAsync(h4, mm, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return TL_ERR_LAUNCH;
  const double x0 = fdec(h4[0]), y0 = fdec(h4[1]), x1 = fdec(h4[2]), y1 = fdec(h4[3]);
  if (!std::isfinite(x0) || !std::isfinite(y0) || !std::isfinite(x1) || !std::isfinite(y1)) return TL_ERR_ARG;   // NaN / inf coordinates
  double ext = std::max(x1 - x0, y1 - y0);
  if (!(ext > 0.0)) ext = 1.0;
  ext *= 1.0 + 1e-9;                                             // the maximum falls inside the last cell
  int L0 = 2;
  while (L0 < 10 && ((int64_t)1 << (2 * L0)) < n) ++L0;          // probe: about one cell per point, at most 1024 x 1024
  Grid g{x0, y0, ext / (double)(1 << L0), 0.0, L0};
  const int64_t nw = ((int64_t)1 << (2 * L0)) / 32 > 0 ? ((int64_t)1 << (2 * L0)) / 32 : 1;
  if (hipMemsetAsync(bits, 0, nw * 4, s) != hipSuccess) return TL_ERR_LAUNCH;
  k_probe<<<tl_grid(n, kBlock), kBlock, 0, s>>>(xy, n, g, bits);
  k_popcount<<<tl_grid(nw, kBlock), kBlock, 0, s>>>(bits, nw, mm + 4);
  unsigned occ = 0;
  if (hipMemcpyAsync(&occ, mm + 4, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return TL_ERR_LAUNCH;
  TL_CHECK_LAUNCH();
  double per_cell = (double)n / (double)(occ ? occ : 1);
  int L = L0;
  while (per_cell > 12.0 && L < kMaxLevel) { per_cell *= 0.25; ++L; }    // one more level quarters a locally uniform cell
  out->lo[0] = x0; out->lo[1] = y0; out->h = ext / (double)(1 << L); out->levels = L; out->reserved = 0;
  return TL_OK;
}

int64_t tl_hdbscan_grid_ws_bytes(int64_t n, const TlHdbGrid* grid) {
  if (n < 2 || !grid_ok(grid)) return 0;
  return carve(nullptr, n, grid->levels).bytes;
}

// workspace for min_samples beyond the 128 neighbours k_core keeps in registers: + one heap of min_samples doubles per walking thread
int64_t tl_hdbscan_grid_ws_bytes_k(int64_t n, const TlHdbGrid* grid, int min_samples) {
  const int64_t base = tl_hdbscan_grid_ws_bytes(n, grid);
  if (!base || min_samples < 1 || min_samples > kBigMaxK) return 0;
  return min_samples > kMaxK ? a16(base) + (int64_t)kBigThreads * min_samples * 8 : base;
}

// xy f32[n,2] (device) -> the n-1 edges of the minimum spanning tree of the mutual-reachability graph (original point indices,
// e_src < e_dst, in no particular order; sort by (weight, src, dst) for a canonical list), core f64[n] or NULL.
// SYNCHRONISES the stream once per Boruvka round (<= ~20) to read the edge count back.
int tl_hdbscan_mst_grid(const float* xy, int64_t n, int min_samples, const TlHdbGrid* grid, int32_t* e_src, int32_t* e_dst, double* e_w,
                        double* core_out, void* ws, tl_stream_t stream) {
  if (!xy || !e_src || !e_dst || !e_w || !ws || n < 2 || n > 0x7FFFFFF0 || min_samples < 1 || min_samples > kBigMaxK || !grid_ok(grid)) return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  const int L = grid->levels, ni = (int)n;
  const Grid g{grid->lo[0], grid->lo[1], grid->h, (std::fabs(grid->lo[0]) + std::fabs(grid->lo[1]) + grid->h * (double)(1 << L)) * 1e-13, L};
  const int64_t ncell = (int64_t)1 << (2 * L);
  const Ws w = carve(ws, n, L);
  const unsigned gp = tl_grid(n, kBlock), gc = tl_grid(ncell, kBlock);
  const unsigned gpt = (unsigned)tl_cdiv(n, kBlock);
  // sort by leaf cell
  if (hipMemsetAsync(w.cnt, 0, ncell * 4, s) != hipSuccess) return TL_ERR_LAUNCH;
  k_count<<<gp, kBlock, 0, s>>>(xy, ni, g, w.cell, w.cnt);
  const int64_t nb = tl_cdiv(ncell, kTile);
  k_scan1<<<(unsigned)nb, kBlock, 0, s>>>(w.cnt, ncell, w.part);
  k_scan2<<<1, kBlock, 0, s>>>(w.part, nb);
  k_scan3<<<(unsigned)nb, kBlock, 0, s>>>(w.cnt, ncell, w.part, w.start);
  if (hipMemsetAsync(w.cnt, 0, ncell * 4, s) != hipSuccess) return TL_ERR_LAUNCH;
  k_scatter<<<gp, kBlock, 0, s>>>(xy, ni, w.cell, w.start, w.cnt, w.sx, w.sy, w.oid);
  auto tags = [&](const int* comp) {
    k_tags_leaf<<<gc, kBlock, 0, s>>>(w.start, ncell, comp, w.tag + lvl_off(L));
    for (int l = L - 1; l >= 0; --l) k_tags_up<<<tl_grid((int64_t)1 << (2 * l), kBlock), kBlock, 0, s>>>(w.tag + lvl_off(l + 1), w.tag + lvl_off(l), l);
  };
  // core distances
  tags(nullptr);
  if (min_samples <= kMaxK) k_core<<<gpt, kBlock, 0, s>>>(w.sx, w.sy, ni, min_samples, g, w.start, w.tag, w.core);
  else                                                          // the heaps sit behind the regular workspace (tl_hdbscan_grid_ws_bytes_k)
    k_core_big<<<kBigThreads / kBlock, kBlock, 0, s>>>(w.sx, w.sy, ni, min_samples, g, w.start, w.tag, w.core, (double*)((char*)ws + a16(w.bytes)));
  k_mincore_leaf<<<gc, kBlock, 0, s>>>(w.start, ncell, w.core, w.mincore + lvl_off(L));
  for (int l = L - 1; l >= 0; --l) k_mincore_up<<<tl_grid((int64_t)1 << (2 * l), kBlock), kBlock, 0, s>>>(w.mincore + lvl_off(l + 1), w.mincore + lvl_off(l), l);
  if (core_out) k_unsort<<<gp, kBlock, 0, s>>>(w.core, w.oid, ni, core_out);
  // Boruvka rounds
  k_init_uf<<<gp, kBlock, 0, s>>>(w.parent, ni);
  if (hipMemsetAsync(w.n_edges, 0, 4, s) != hipSuccess) return TL_ERR_LAUNCH;
  int edges = 0;
  for (int round = 0; edges < ni - 1; ++round) {
    if (round >= 64) return TL_ERR_LAUNCH;
    k_flatten<<<gp, kBlock, 0, s>>>(w.parent, w.comp, ni, w.comp_best, w.win);
    tags(w.comp);
    k_search<<<gpt, kBlock, 0, s>>>(w.sx, w.sy, w.core, w.oid, w.comp, ni, g, w.start, w.tag, w.mincore, w.comp_best, w.pv, w.pj);
    k_win<<<gp, kBlock, 0, s>>>(w.pv, w.pj, w.oid, w.comp, ni, w.comp_best, w.win);
    k_merge<<<gp, kBlock, 0, s>>>(w.pv, w.pj, w.oid, w.comp, ni, w.comp_best, w.win, w.parent, w.n_edges, e_src, e_dst, e_w);
    int now = 0;
    if (hipMemcpyAsync(&now, w.n_edges, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return TL_ERR_LAUNCH;
    if (now <= edges) return TL_ERR_LAUNCH;                      // a round must merge something (non-finite coordinates would not)
    edges = now;
  }
  TL_CHECK_LAUNCH();
  return TL_OK;
}

}  // extern "C"
