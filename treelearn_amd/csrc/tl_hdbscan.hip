// HDBSCAN grouping (reference tree_learn/util/pipeline.py:184-191: sklearn HDBSCAN(min_cluster_size=m),
// i.e. min_samples = m, euclidean, Prim MST on mutual reachability, EOM selection).
//
// Device part (this file, tl_hdbscan_mst): exact restatement of sklearn's two O(n^2) stages in fp64 on the
// fp32 inputs -- k-th-nearest-neighbour core distances (sklearn _hdbscan/hdbscan.py:278-357) and Prim's
// algorithm over the complete mutual-reachability graph with sklearn's tie-breaking (smallest index among
// equal reachabilities, sources replaced only on strict improvement; sklearn _hdbscan/_linkage.pyx:111-176).
// sklearn runs the Prim loop single-threaded; here every step is one launch over all candidates.
// Host part (tl_hdbscan_labels_host): sort edges, single-linkage tree, condensed tree, stabilities, EOM,
// labelling (sklearn _hdbscan/_linkage.pyx make_single_linkage, _tree.pyx) -- sequential pointer chasing over
// n-1 merges, done on the host like the reference does.
#include "tl_common.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <map>
#include <numeric>
#include <set>
#include <vector>

namespace {

constexpr int kBlock = 256;
constexpr int kMaxK = 128;

// d^2 without fused multiply-add (sklearn's compiled distance loops do not contract)
__device__ __forceinline__ double dist2(double ax, double ay, double bx, double by) {
  const double dx = __dsub_rn(ax, bx), dy = __dsub_rn(ay, by);
  return __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
}

// core[i] = distance to the k-th nearest point, the point itself included (kneighbors(X, k)[:, -1])
__global__ void __launch_bounds__(kBlock) k_core_dist(const float* __restrict__ xy, int n, int k, double* __restrict__ core) {
  __shared__ double tx[kBlock], ty[kBlock];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const double x = i < n ? (double)xy[2 * (int64_t)i] : 0.0, y = i < n ? (double)xy[2 * (int64_t)i + 1] : 0.0;
  double best[kMaxK];                                   // k smallest d^2 so far (unsorted), worst tracked
  int cnt = 0; double worst = -1.0; int worst_at = 0;
  for (int j0 = 0; j0 < n; j0 += kBlock) {
    const int j = j0 + threadIdx.x;
    __syncthreads();
    tx[threadIdx.x] = j < n ? (double)xy[2 * (int64_t)j] : 0.0;
    ty[threadIdx.x] = j < n ? (double)xy[2 * (int64_t)j + 1] : 0.0;
    __syncthreads();
    if (i < n) {
      const int m = min(kBlock, n - j0);
      for (int t = 0; t < m; ++t) {
        const double d = dist2(x, y, tx[t], ty[t]);
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
  }
  if (i < n) core[i] = sqrt(cnt >= k ? worst : std::numeric_limits<double>::infinity());
}

struct Cand { double val; int node; int src; };

__device__ __forceinline__ bool better(double v, int j, double bv, int bj) { return v < bv || (v == bv && j < bj); }

// One Prim step.  Every block first reduces the previous step's per-block candidates to learn the node that
// just joined the tree (block 0 also records the edge), then relaxes its share of the remaining nodes against
// it and publishes its best candidate for the next step.
__global__ void __launch_bounds__(kBlock) k_prim_step(const float* __restrict__ xy, const double* __restrict__ core, int n, int step,
                                                      double* __restrict__ reach, int* __restrict__ source, unsigned char* __restrict__ in_tree,
                                                      Cand* __restrict__ cand /*[2][gridDim.x]*/, int* __restrict__ e_src, int* __restrict__ e_dst,
                                                      double* __restrict__ e_w) {
  __shared__ double sv[kBlock]; __shared__ int sj[kBlock]; __shared__ int ss[kBlock];
  const int nb = gridDim.x;
  int cur = 0;
  if (step > 0) {
    const Cand* prev = cand + ((step - 1) & 1) * nb;
    double bv = std::numeric_limits<double>::infinity(); int bj = 0x7FFFFFFF, bs = 0;
    for (int b = threadIdx.x; b < nb; b += kBlock) { const Cand c = prev[b]; if (better(c.val, c.node, bv, bj)) { bv = c.val; bj = c.node; bs = c.src; } }
    sv[threadIdx.x] = bv; sj[threadIdx.x] = bj; ss[threadIdx.x] = bs;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
      if (threadIdx.x < off && better(sv[threadIdx.x + off], sj[threadIdx.x + off], sv[threadIdx.x], sj[threadIdx.x])) {
        sv[threadIdx.x] = sv[threadIdx.x + off]; sj[threadIdx.x] = sj[threadIdx.x + off]; ss[threadIdx.x] = ss[threadIdx.x + off];
      }
      __syncthreads();
    }
    cur = sj[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) { e_src[step - 1] = ss[0]; e_dst[step - 1] = cur; e_w[step - 1] = sv[0]; in_tree[cur] = 1; }
    __syncthreads();
  } else if (blockIdx.x == 0 && threadIdx.x == 0) {
    in_tree[0] = 1;
  }
  if (step >= n - 1) return;                             // final call only records the last edge
  const double cx = (double)xy[2 * (int64_t)cur], cy = (double)xy[2 * (int64_t)cur + 1], cc = core[cur];
  double bv = std::numeric_limits<double>::infinity(); int bj = 0x7FFFFFFF, bs = 0;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += nb * kBlock) {
    if (j == cur || in_tree[j]) continue;
    const double d = sqrt(dist2(cx, cy, (double)xy[2 * (int64_t)j], (double)xy[2 * (int64_t)j + 1]));
    const double mr = fmax(fmax(cc, core[j]), d);
    double r = reach[j]; int s = source[j];
    if (mr < r) { r = mr; s = cur; reach[j] = r; source[j] = s; }
    if (better(r, j, bv, bj)) { bv = r; bj = j; bs = s; }
  }
  sv[threadIdx.x] = bv; sj[threadIdx.x] = bj; ss[threadIdx.x] = bs;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off && better(sv[threadIdx.x + off], sj[threadIdx.x + off], sv[threadIdx.x], sj[threadIdx.x])) {
      sv[threadIdx.x] = sv[threadIdx.x + off]; sj[threadIdx.x] = sj[threadIdx.x + off]; ss[threadIdx.x] = ss[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { Cand c; c.val = sv[0]; c.node = sj[0]; c.src = ss[0]; cand[(step & 1) * nb + blockIdx.x] = c; }
}

__global__ void k_prim_init(int n, double* reach, int* source, unsigned char* in_tree) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    reach[i] = std::numeric_limits<double>::infinity(); source[i] = 1; in_tree[i] = 0;
  }
}

inline int prim_blocks(int64_t n) { int64_t b = tl_cdiv(n, kBlock * 4); return (int)(b < 1 ? 1 : (b > 256 ? 256 : b)); }
inline int64_t a16(int64_t x) { return (x + 15) & ~(int64_t)15; }

}  // namespace

extern "C" {

int64_t tl_hdbscan_ws_bytes(int64_t n) {
  if (n <= 0) return 0;
  return a16(n * 8) /*core*/ + a16(n * 8) /*reach*/ + a16(n * 4) /*source*/ + a16(n) /*in_tree*/ + a16(2 * 256 * (int64_t)sizeof(Cand)) + 64;
}

// xy f32[n,2] (device).  Outputs (device): e_src i32[n-1], e_dst i32[n-1], e_w f64[n-1] = the MST edges in
// the order Prim adds them; core f64[n] may be NULL.  ws: tl_hdbscan_ws_bytes(n).
int tl_hdbscan_mst(const float* xy, int64_t n, int min_samples, int32_t* e_src, int32_t* e_dst, double* e_w, double* core_out,
                   void* ws, tl_stream_t stream) {
  if (!xy || !e_src || !e_dst || !e_w || !ws || n < 2 || n > 0x7FFFFFF0 || min_samples < 1 || min_samples > kMaxK) return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  char* p = (char*)ws;
  double* core = (double*)p; p += a16(n * 8);
  double* reach = (double*)p; p += a16(n * 8);
  int* source = (int*)p; p += a16(n * 4);
  unsigned char* in_tree = (unsigned char*)p; p += a16(n);
  Cand* cand = (Cand*)p;
  const int ni = (int)n;
  k_core_dist<<<(unsigned)tl_cdiv(n, kBlock), kBlock, 0, s>>>(xy, ni, min_samples, core);
  k_prim_init<<<tl_grid(n, kBlock), kBlock, 0, s>>>(ni, reach, source, in_tree);
  const int nb = prim_blocks(n);
  for (int step = 0; step <= ni - 1; ++step)
    k_prim_step<<<nb, kBlock, 0, s>>>(xy, core, ni, step, reach, source, in_tree, cand, e_src, e_dst, e_w);
  if (core_out && hipMemcpyAsync(core_out, core, n * 8, hipMemcpyDeviceToDevice, s) != hipSuccess) return TL_ERR_LAUNCH;
  TL_CHECK_LAUNCH();
  return TL_OK;
}

// HOST function: re-orders a spanning tree (edges in any order and orientation) the way Prim's algorithm started at point 0 walks it
// -- lightest frontier edge first, smallest new point index among equal weights, src = the end already in the tree -- i.e. the
// order and orientation sklearn's Prim (and tl_hdbscan_mst) emits when the tree is the one it finds.  tl_hdbscan_labels_host breaks
// weight ties by position and numbers clusters by orientation, so this makes the labels of a tree from tl_hdbscan_mst_grid agree
// with the Prim form wherever equal weights do not make the trees themselves differ.
int tl_hdbscan_prim_order_host(const int32_t* e_src, const int32_t* e_dst, const double* e_w, int64_t n, int32_t* o_src, int32_t* o_dst, double* o_w) {
  if (!e_src || !e_dst || !e_w || !o_src || !o_dst || !o_w || n < 2) return TL_ERR_ARG;
  const int64_t m = n - 1;
  std::vector<int64_t> off(n + 1, 0);
  for (int64_t e = 0; e < m; ++e) {
    if (e_src[e] < 0 || e_src[e] >= n || e_dst[e] < 0 || e_dst[e] >= n) return TL_ERR_ARG;
    ++off[e_src[e] + 1]; ++off[e_dst[e] + 1];
  }
  for (int64_t i = 0; i < n; ++i) off[i + 1] += off[i];
  std::vector<int64_t> adj(2 * m), cur(off.begin(), off.end() - 1);
  for (int64_t e = 0; e < m; ++e) { adj[cur[e_src[e]]++] = e; adj[cur[e_dst[e]]++] = e; }
  struct Item { double w; int32_t v, u; };
  auto later = [](const Item& a, const Item& b) { return a.w > b.w || (a.w == b.w && a.v > b.v); };
  std::vector<Item> heap;
  heap.reserve(n);
  std::vector<char> in_tree(n, 0);
  auto grow = [&](int32_t u) {
    in_tree[u] = 1;
    for (int64_t q = off[u]; q < off[u + 1]; ++q) {
      const int64_t e = adj[q];
      const int32_t v = e_src[e] == u ? e_dst[e] : e_src[e];
      if (!in_tree[v]) { heap.push_back({e_w[e], v, u}); std::push_heap(heap.begin(), heap.end(), later); }
    }
  };
  grow(0);
  int64_t k = 0;
  while (!heap.empty()) {
    std::pop_heap(heap.begin(), heap.end(), later);
    const Item it = heap.back(); heap.pop_back();
    if (in_tree[it.v]) continue;
    if (k >= m) return TL_ERR_ARG;
    o_src[k] = it.u; o_dst[k] = it.v; o_w[k] = it.w; ++k;
    grow(it.v);
  }
  return k == m ? TL_OK : TL_ERR_ARG;                          // k < m: the edges do not span the points
}

// HOST function (host pointers, no GPU work): MST edges -> HDBSCAN labels (-1 = noise, clusters 0..K-1 in
// ascending condensed-tree id, as sklearn numbers them).  EOM selection, allow_single_cluster = False,
// cluster_selection_epsilon = 0.
int tl_hdbscan_labels_host(const int32_t* e_src, const int32_t* e_dst, const double* e_w, int64_t n, int min_cluster_size, int32_t* labels) {
  if (!e_src || !e_dst || !e_w || !labels || n < 2 || min_cluster_size < 2) return TL_ERR_ARG;
  const int64_t m = n - 1;
  // --- _process_mst: edges by ascending weight (stable), make_single_linkage
  std::vector<int64_t> order(m);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return e_w[a] < e_w[b]; });
  std::vector<int64_t> uf_parent(2 * n - 1, -1), uf_size(2 * n - 1, 0);
  for (int64_t i = 0; i < n; ++i) uf_size[i] = 1;
  auto fast_find = [&](int64_t x) {
    int64_t r = x;
    while (uf_parent[r] != -1) r = uf_parent[r];
    while (uf_parent[x] != -1 && uf_parent[x] != r) { const int64_t nx = uf_parent[x]; uf_parent[x] = r; x = nx; }
    return r;
  };
  std::vector<int64_t> left(m), right(m), csize(m);
  std::vector<double> value(m);
  int64_t next_label = n;
  for (int64_t i = 0; i < m; ++i) {
    const int64_t e = order[i];
    const int64_t a = fast_find(e_src[e]), b = fast_find(e_dst[e]);
    left[i] = a; right[i] = b; value[i] = e_w[e]; csize[i] = uf_size[a] + uf_size[b];
    uf_parent[a] = next_label; uf_parent[b] = next_label; uf_size[next_label] = csize[i]; ++next_label;
  }
  // --- _condense_tree
  const int64_t root = 2 * (n - 1);
  // breadth-first list of the nodes below `start` (level by level, left before right -- the order sklearn's bfs_from_hierarchy
  // gives); the output vector doubles as the queue, so the ~n calls on small sub-trees allocate nothing
  auto bfs = [&](int64_t start, std::vector<int64_t>& out) {
    out.clear();
    out.push_back(start);
    for (size_t head = 0; head < out.size(); ++head) {
      const int64_t x = out[head];
      if (x >= n) { out.push_back(left[x - n]); out.push_back(right[x - n]); }
    }
  };
  struct Row { int64_t parent, child; double lambda; int64_t size; };
  std::vector<Row> tree;
  tree.reserve(2 * n);
  std::vector<int64_t> relabel(root + 1, 0), node_list, sub;
  std::vector<char> ignore(root + 1, 0);
  relabel[root] = n;
  int64_t next_cluster = n + 1;
  bfs(root, node_list);
  const double INF = std::numeric_limits<double>::infinity();
  for (int64_t node : node_list) {
    if (ignore[node] || node < n) continue;
    const int64_t l = left[node - n], r = right[node - n];
    const double dist = value[node - n];
    const double lambda = dist > 0.0 ? 1.0 / dist : INF;
    const int64_t lc = l >= n ? csize[l - n] : 1, rc = r >= n ? csize[r - n] : 1;
    auto spill = [&](int64_t from) {
      bfs(from, sub);
      for (int64_t sn : sub) { if (sn < n) tree.push_back({relabel[node], sn, lambda, 1}); ignore[sn] = 1; }
    };
    if (lc >= min_cluster_size && rc >= min_cluster_size) {
      relabel[l] = next_cluster++; tree.push_back({relabel[node], relabel[l], lambda, lc});
      relabel[r] = next_cluster++; tree.push_back({relabel[node], relabel[r], lambda, rc});
    } else if (lc < min_cluster_size && rc < min_cluster_size) {
      spill(l); spill(r);
    } else if (lc < min_cluster_size) {
      relabel[r] = relabel[node]; spill(l);
    } else {
      relabel[l] = relabel[node]; spill(r);
    }
  }
  // --- _compute_stability
  int64_t largest_child = 0, smallest_cluster = std::numeric_limits<int64_t>::max(), largest_parent = 0;
  for (const Row& t : tree) { largest_child = std::max(largest_child, t.child); smallest_cluster = std::min(smallest_cluster, t.parent); largest_parent = std::max(largest_parent, t.parent); }
  largest_child = std::max(largest_child, smallest_cluster);
  const int64_t num_clusters = largest_parent - smallest_cluster + 1;
  std::vector<double> births(largest_child + 1, std::nan(""));
  for (const Row& t : tree) births[t.child] = t.lambda;
  births[smallest_cluster] = 0.0;
  std::vector<double> stab(num_clusters, 0.0);
  for (const Row& t : tree) stab[t.parent - smallest_cluster] += (t.lambda - births[t.parent]) * (double)t.size;
  // --- _get_clusters (EOM), root excluded
  std::vector<std::vector<int64_t>> kids(num_clusters);          // cluster tree (rows with size > 1)
  for (const Row& t : tree) if (t.size > 1) kids[t.parent - smallest_cluster].push_back(t.child);
  std::vector<char> is_cluster(num_clusters, 1);
  is_cluster[0] = 0;                                             // the root is never a cluster (allow_single_cluster=False)
  for (int64_t c = largest_parent; c > smallest_cluster; --c) {  // node_list = sorted(keys, reverse)[:-1]
    const int64_t ci = c - smallest_cluster;
    double sub_stab = 0.0;
    for (int64_t k : kids[ci]) sub_stab += stab[k - smallest_cluster];
    if (sub_stab > stab[ci]) {
      is_cluster[ci] = 0; stab[ci] = sub_stab;
    } else {
      std::vector<int64_t> stack(kids[ci].begin(), kids[ci].end());
      while (!stack.empty()) {
        const int64_t x = stack.back(); stack.pop_back();
        is_cluster[x - smallest_cluster] = 0;
        for (int64_t k : kids[x - smallest_cluster]) stack.push_back(k);
      }
    }
  }
  std::vector<int64_t> label_of(num_clusters, -1);
  int64_t nlab = 0;
  for (int64_t ci = 0; ci < num_clusters; ++ci) if (is_cluster[ci]) label_of[ci] = nlab++;
  // --- _do_labelling: a point belongs to the first SELECTED cluster on its way up the condensed tree (the
  // reference unions every non-selected child into its parent and looks up the representative), else noise
  std::vector<int64_t> par(largest_parent + 1, -1);
  for (const Row& t : tree) par[t.child] = t.parent;
  for (int64_t i = 0; i < n; ++i) {
    int64_t x = i;
    while (true) {
      if (x >= smallest_cluster && (is_cluster[x - smallest_cluster] || x == smallest_cluster)) break;
      if (par[x] < 0) { x = smallest_cluster; break; }
      x = par[x];
    }
    labels[i] = (x == smallest_cluster) ? -1 : (int32_t)label_of[x - smallest_cluster];
  }
  return TL_OK;
}

}  // extern "C"
