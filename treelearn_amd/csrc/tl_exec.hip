// tl_exec.hip -- tl_forward: the whole eval-mode forward of one batch of tiles enqueued from C.
//
// What the reference's tile loop calls once per tile, `model(batch, return_loss=False)` (tree_learn/util/pipeline.py:86 ->
// tree_learn/model/tree_learn.py:75-103), is here ~80 launches: voxel coordinates, the bitmap pyramid, the block-local order of level 1,
// every rulebook, 72 sparse convs in the pre-activated dataflow (BatchNorm + ReLU in producer epilogues or applied at staging, residual
// adds and the skip concat as views; blocks.py:55-79,137-149) and the fused heads.  Driven from Python that is ~80 ctypes calls with ~60
// field stores each plus a tensor allocation per launch; here it is one call: the layout arithmetic between the two read-backs takes
// microseconds, activations come out of the caller's arena through a best-fit free list (planned by a dry run of the same code, so
// `needed_bytes` is exact), and the launch thread never waits for an interpreter.  Same kernels, same order, same arguments as
// treelearn_amd/model/engine.py issues them: results are bit-identical (tests/test_gpu_exec.py).
#include <string.h>
#include <vector>

#include "tl_common.h"
#include "tl_arena.h"

namespace {

constexpr int64_t kCompactMinRows = 65536;      // geometry.COMPACT_MIN_ROWS: levels from this size on also get the column form of their rulebook
constexpr int64_t kBlkMinRows = 16384;          // geometry.BLK_MIN_ROWS / BLK_MAX_ROWS: the level-1 sizes that go into the block-local order
constexpr int64_t kBlkMaxRows = (1 << 25) - 64;
constexpr size_t kHostBytes = 64 + (size_t)TL_POINT_COORDS_MAX_PARTS * 16;

struct Ten {                                     // an activation matrix [n, C] (or a column view of a wider one)
  int64_t off = -1;                              // arena offset of the element [0, 0]
  int64_t ld = 0, n = 0;
  int C = 0;
  int64_t own_off = -1, own_bytes = 0;           // the allocation to return, -1 = a view / not owned
};

struct ViewReq {                                 // one requested view of a conv result (engine.View)
  const Ten* dst = nullptr;                      // destination (column view) or nullptr = allocate
  tl_affine aff{nullptr, nullptr};
  int relu = 0;
};

struct Table {
  const int32_t* table = nullptr;
  const int32_t* compact = nullptr;
  const int32_t* scatter = nullptr;
  bool blk = false;
  int one_hot = 0;
};

struct LevelG {
  int32_t dims[4];
  int64_t n = 0;
  int64_t coords = -1, nbr = -1, ct = -1, child = -1, inv = -1, invp = -1;       // word offsets into the geometry block, -1 = absent (no `parent`: the forward never reads it)
};

}  // namespace

struct tl_exec {
  int32_t* host = nullptr;                       // pinned: 16 words of read-back + TL_POINT_COORDS_MAX_PARTS rows of per-workgroup extents
  hipEvent_t ev_main = nullptr, ev_side = nullptr, ev_flag = nullptr;
  bool flag_in_flight = false;                   // a unit-builder flag read-back has been enqueued and not been waited for yet
  bool profile = false;
  std::vector<hipEvent_t> pev;                   // event pairs of the profiled forward
  std::vector<tl_launch_rec> recs;
  hipStream_t prof_stream = nullptr;
};

namespace {

struct Run {
  tl_exec* ex;
  const tl_net_desc* net;
  tl_forward_args* a;
  hipStream_t s;
  Arena ar;
  int esize;
  int rc = TL_OK;
  int launches = 0;
  int nl;
  LevelG lv[TL_MAX_LEVELS];
  bool blocked = false;
  int64_t g0 = 0;                                // arena offset of the geometry block (word offsets are relative to it)
  int64_t o_unit = -1, o_counter = -1, o_halo = -1, o_lrb = -1, o_pmask = -1, o_nn = -1;

  const int32_t* gw(int64_t word_off) const { return word_off < 0 ? nullptr : reinterpret_cast<const int32_t*>(ar.at(g0 + 4 * word_off)); }

  Ten alloc(int64_t n, int C) {
    Ten t;
    t.n = n; t.C = C; t.ld = C; t.own_bytes = n * C * esize; t.own_off = t.off = ar.take(t.own_bytes);
    return t;
  }
  void release(Ten& t) {
    if (t.own_off >= 0) ar.give(t.own_off, t.own_bytes);
    t.own_off = -1;
  }
  Ten cols(const Ten& t, int c0, int C) const {
    Ten v;
    v.off = t.off + (int64_t)c0 * esize; v.ld = t.ld; v.n = t.n; v.C = C;
    return v;
  }

  Table subm(int li) const {
    Table t;
    if (li == 0 && blocked) { t.blk = true; return t; }
    t.table = gw(lv[li].nbr); t.compact = gw(lv[li].ct);
    return t;
  }
  Table down(int li) const { Table t; t.table = gw(lv[li].child); return t; }
  Table up(int li) const {
    Table t;
    t.scatter = gw(lv[li].child);
    if (lv[li].invp >= 0) { t.table = gw(lv[li].invp); t.one_hot = 2; }       // packed: (parent << 3) | tap per row
    else { t.table = gw(lv[li].inv); t.one_hot = 1; }
    return t;
  }

  // one tl_conv_fwd launch, arguments as ops.conv_fwd fills them; `outs` receives the result matrices in the order of `views`
  void conv(int level, int kind, const Ten& x, const tl_weight& w, const Table& tb, int64_t n_out, const ViewReq* views, int nviews, Ten* outs,
            const Ten* residual = nullptr, tl_affine in_aff = {nullptr, nullptr}, int in_relu = 0, int all_ones = 0, int split_part = -1, int split_cin = 0) {
    for (int i = 0; i < nviews; ++i) outs[i] = views[i].dst ? *views[i].dst : alloc(n_out, w.Cout);
    if (rc != TL_OK) return;
    ++launches;
    if (ar.dry) return;
    tl_conv_args c{};
    c.in = ar.at(x.off); c.in_ld = x.ld; c.weight = w.w; c.weight_frag = w.frag; c.weight_x3 = w.x3;
    c.table = tb.table; c.table_compact = tb.compact; c.table_one_hot = tb.one_hot; c.table_scatter = tb.scatter;
    if (tb.blk) {
      c.blk_unit = gw(o_unit); c.blk_counter = gw(o_counter); c.blk_halo = gw(o_halo);
      c.blk_lrb = reinterpret_cast<const uint32_t*>(gw(o_lrb)); c.blk_pmask = gw(o_pmask);
    }
    c.in_all_ones = all_ones;
    c.n_out = n_out; c.n_in = x.n; c.K = w.K; c.Cin = w.Cin; c.Cout = w.Cout; c.dtype = net->dtype;
    c.in_scale = in_aff.scale; c.in_shift = in_aff.shift; c.in_relu = in_relu;
    if (residual) { c.residual = ar.at(residual->off); c.res_ld = residual->ld; }
    c.out = ar.at(outs[0].off); c.out_ld = outs[0].ld; c.out_scale = views[0].aff.scale; c.out_shift = views[0].aff.shift; c.out_relu = views[0].relu;
    if (nviews > 1) { c.out2 = ar.at(outs[1].off); c.out2_ld = outs[1].ld; c.out2_scale = views[1].aff.scale; c.out2_shift = views[1].aff.shift; c.out2_relu = views[1].relu; }
    if (nviews > 2) { c.out3 = ar.at(outs[2].off); c.out3_ld = outs[2].ld; c.out3_scale = views[2].aff.scale; c.out3_shift = views[2].aff.shift; c.out3_relu = views[2].relu; }
    const bool prof = ex->profile;
    if (prof) {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = TL_ERR_LAUNCH; return; }
      ex->pev.push_back(e0); ex->pev.push_back(e1);
      tl_launch_rec r{};
      r.level = level; r.kind = kind; r.K = w.K; r.Cin = w.Cin; r.Cout = w.Cout; r.residual = residual != nullptr; r.esize = esize;
      r.split_part = split_part; r.split_cin = split_cin; r.in_prologue = in_aff.scale != nullptr || in_relu; r.n_out = n_out; r.n_in = x.n;
      ex->recs.push_back(r);
      if (hipEventRecord(e0, s) != hipSuccess) rc = TL_ERR_LAUNCH;
    }
    const int r = tl_conv_fwd(&c, s);
    if (prof && hipEventRecord(ex->pev.back(), s) != hipSuccess) rc = TL_ERR_LAUNCH;
    if (r != TL_OK) rc = r;
  }

  static ViewReq RAW(const Ten* dst = nullptr) { ViewReq v; v.dst = dst; return v; }
  static ViewReq ACT(const tl_affine& aff, const Ten* dst = nullptr) { ViewReq v; v.dst = dst; v.aff = aff; v.relu = 1; return v; }

  // engine._Res.run: t = conv1(x_act) [epilogue bn3 + relu]; y = conv2(t) + i_branch(x_raw)     (blocks.py:55-79)
  void res_block(int li, const tl_res_desc& b, const Ten& x_raw, const Ten& x_act, const ViewReq* views, int nviews, Ten* outs) {
    const Table nb = subm(li);
    const int64_t n = lv[li].n;
    Ten t, res;
    const ViewReq v1 = ACT(b.bn3);
    conv(li, 0, x_act, b.w1, nb, n, &v1, 1, &t);
    const bool has1x1 = b.w1x1.w != nullptr;
    if (has1x1) { const ViewReq vr = RAW(); conv(li, 3, x_raw, b.w1x1, Table{}, n, &vr, 1, &res); }
    conv(li, 0, t, b.w2, nb, n, views, nviews, outs, has1x1 ? &res : &x_raw);
    release(t);
    if (has1x1) release(res);
  }

  // engine._U.run (takes over x_raw / x_act)      (blocks.py:137-149)
  void ublock(int li, Ten x_raw, Ten x_act, const ViewReq* views, int nviews, Ten* outs) {
    const tl_ublock_desc& u = net->u[li];
    const int64_t n = lv[li].n;
    const int C = u.C;
    Ten o2[3];
    { const ViewReq v[2] = {RAW(), ACT(u.blocks[1].bn0)}; res_block(li, u.blocks[0], x_raw, x_act, v, 2, o2); }
    release(x_raw); release(x_act);
    x_raw = o2[0]; x_act = o2[1];
    if (!u.deeper) {
      res_block(li, u.blocks[1], x_raw, x_act, views, nviews, outs);
      release(x_raw); release(x_act);
      return;
    }
    Ten cat_raw = alloc(n, 2 * C), cat_act = alloc(n, 2 * C);
    const Ten cr_l = cols(cat_raw, 0, C), cr_r = cols(cat_raw, C, C), ca_l = cols(cat_act, 0, C), ca_r = cols(cat_act, C, C);
    Ten o3[3];
    { const ViewReq v[3] = {RAW(&cr_l), ACT(u.bn_cat_l, &ca_l), ACT(u.bn_down)}; res_block(li, u.blocks[1], x_raw, x_act, v, 3, o3); }
    release(x_raw); release(x_act);
    Ten xd = o3[2], d[2];
    { const ViewReq v[2] = {RAW(), ACT(net->u[li + 1].blocks[0].bn0)}; conv(li, 1, xd, u.wd, down(li), lv[li + 1].n, v, 2, d); }
    release(xd);
    Ten e_act;
    { const ViewReq v = ACT(u.bn_up); ublock(li + 1, d[0], d[1], &v, 1, &e_act); }
    { const ViewReq v[2] = {RAW(&cr_r), ACT(u.bn_cat_r, &ca_r)}; Ten o[2]; conv(li, 2, e_act, u.wu, up(li), n, v, 2, o); }
    release(e_act);
    Ten y[2];
    { const ViewReq v[2] = {RAW(), ACT(u.tail[1].bn0)}; res_block(li, u.tail[0], cat_raw, cat_act, v, 2, y); }
    release(cat_raw); release(cat_act);
    res_block(li, u.tail[1], y[0], y[1], views, nviews, outs);
    release(y[0]); release(y[1]);
  }

  // engine._U.run_l1_staged: level 1 on block-local rows, BatchNorm + ReLU of every block's first conv applied at staging
  void l1_block(const tl_res_desc& b, const Ten& x, const tl_affine& in_aff, const ViewReq* views, int nviews, Ten* outs) {
    const Table nb = subm(0);
    const int64_t n = lv[0].n;
    Ten t, res;
    const ViewReq v1 = ACT(b.bn3);
    conv(0, 0, x, b.w1, nb, n, &v1, 1, &t, nullptr, in_aff, 1);
    const bool has1x1 = b.w1x1.w != nullptr;
    if (has1x1) { const ViewReq vr = RAW(); conv(0, 3, x, b.w1x1, Table{}, n, &vr, 1, &res); }
    conv(0, 0, t, b.w2, nb, n, views, nviews, outs, has1x1 ? &res : &x);
    release(t);
    if (has1x1) release(res);
  }
  void l1_staged(Ten x_raw, const ViewReq* views, int nviews, Ten* outs) {
    const tl_ublock_desc& u = net->u[0];
    const int64_t n = lv[0].n;
    const int C = u.C;
    const Table nb = subm(0);
    Ten x1;
    { const ViewReq v = RAW(); l1_block(u.blocks[0], x_raw, u.blocks[0].bn0, &v, 1, &x1); }
    release(x_raw);
    Ten cat_raw = alloc(n, 2 * C);
    const Ten cr_l = cols(cat_raw, 0, C), cr_r = cols(cat_raw, C, C);
    Ten o2[2];
    { const ViewReq v[2] = {RAW(&cr_l), ACT(u.bn_down)}; l1_block(u.blocks[1], x1, u.blocks[1].bn0, v, 2, o2); }
    release(x1);
    Ten xd = o2[1], d[2];
    { const ViewReq v[2] = {RAW(), ACT(net->u[1].blocks[0].bn0)}; conv(0, 1, xd, u.wd, down(0), lv[1].n, v, 2, d); }
    release(xd);
    Ten e_act;
    { const ViewReq v = ACT(u.bn_up); ublock(1, d[0], d[1], &v, 1, &e_act); }
    { const ViewReq v = RAW(&cr_r); Ten o; conv(0, 2, e_act, u.wu, up(0), n, &v, 1, &o); }
    release(e_act);
    const tl_res_desc& b = u.tail[0];                       // 2C -> C as its two input-channel halves, each with its slice of the BatchNorm
    Ten part, t, res, y;
    { const ViewReq v = RAW(); conv(0, 0, cr_l, b.w1_half[0], nb, n, &v, 1, &part, nullptr, u.bn_cat_l, 1, 0, 0, 2 * C); }
    { const ViewReq v = ACT(b.bn3); conv(0, 0, cr_r, b.w1_half[1], nb, n, &v, 1, &t, &part, u.bn_cat_r, 1, 0, 1, 2 * C); }
    release(part);
    { const ViewReq v = RAW(); conv(0, 3, cat_raw, b.w1x1, Table{}, n, &v, 1, &res); }
    { const ViewReq v = RAW(); conv(0, 0, t, b.w2, nb, n, &v, 1, &y, &res); }
    release(t); release(res); release(cat_raw);
    l1_block(u.tail[1], y, u.tail[1].bn0, views, nviews, outs);
    release(y);
  }

  // InferencePlan.run: input conv, U-Net, heads.  The input conv's operand [n1, in_channels] is the ones matrix (default configuration: the
  // conv then needs no gather) or the voxel-mean features in the level's row order (tl_voxel_feats through this geometry's v2p)
  void network() {
    const int64_t n1 = lv[0].n;
    const bool feats = net->use_coords || net->use_feats;
    Ten ones = alloc(n1, net->in_channels);
    if (feats) {
      const int64_t ws_bytes = 4 * n1 * net->max_points_per_voxel, o_ws = ar.take(ws_bytes);
      if (!ar.dry) {
        const int r = tl_voxel_feats(a->xyz, a->point_feats, net->in_channels - 3, reinterpret_cast<const int64_t*>(gw(o_v2p)), a->N, n1, net->max_points_per_voxel,
                                     net->use_coords, net->use_feats, net->dtype, reinterpret_cast<int32_t*>(ar.at(o_ws)), ar.at(ones.off), s);
        if (r != TL_OK) rc = r;
      }
      ar.give(o_ws, ws_bytes);
    } else if (!ar.dry) {
      hipError_t e;
      if (net->dtype == TL_F32) e = hipMemsetD32Async((hipDeviceptr_t)ar.at(ones.off), 0x3F800000, n1 * net->in_channels, s);
      else e = hipMemsetD16Async((hipDeviceptr_t)ar.at(ones.off), net->dtype == TL_BF16 ? 0x3F80 : 0x3C00, n1 * net->in_channels, s);
      if (e != hipSuccess) rc = TL_ERR_LAUNCH;
    }
    Ten xs[2], x;
    const ViewReq out = RAW();
    if (blocked) {
      const ViewReq v = RAW();
      Table tin = subm(0);
      if (feats) { tin = Table{}; tin.table = gw(o_nn); }     // block-local rows, real features: the plain table in the new order (row-wise the same sums)
      conv(0, 4, ones, net->w_in, tin, n1, &v, 1, xs, nullptr, {nullptr, nullptr}, 0, feats ? 0 : 1);
      release(ones);
      l1_staged(xs[0], &out, 1, &x);
    } else {
      const ViewReq v[2] = {RAW(), ACT(net->u[0].blocks[0].bn0)};
      conv(0, 4, ones, net->w_in, subm(0), n1, v, 2, xs, nullptr, {nullptr, nullptr}, 0, feats ? 0 : 1);
      release(ones);
      ublock(0, xs[0], xs[1], &out, 1, &x);
    }
    if (rc == TL_OK && !ar.dry) {
      const int r = tl_head_mlp(ar.at(x.off), x.ld, net->dtype, x.C, reinterpret_cast<const int64_t*>(gw(o_v2p)), a->N, net->out_bn.scale, net->out_bn.shift,
                                net->head_w1, net->head_b1, net->head_w2, net->head_b2, a->backbone, a->logits, a->offsets, s);
      if (r != TL_OK) rc = r;
    }
    release(x);
  }
  int64_t o_v2p = -1;
};

inline int64_t al64(int64_t w) { return (w + 63) & ~int64_t(63); }

inline bool weight_is(const tl_weight& w, int K, int Cin, int Cout) { return w.w && w.K == K && w.Cin == Cin && w.Cout == Cout; }
inline bool affine_ok(const tl_affine& a) { return a.scale && a.shift; }

// ResidualBlock Cin -> C (blocks.py:42-79): two 27-tap convs, a 1x1 i_branch exactly when the widths differ
inline bool res_ok(const tl_res_desc& b, int Cin, int C) {
  if (!affine_ok(b.bn0) || !affine_ok(b.bn3) || !weight_is(b.w1, 27, Cin, C) || !weight_is(b.w2, 27, C, C)) return false;
  if (Cin != C ? !weight_is(b.w1x1, 1, Cin, C) : b.w1x1.w != nullptr) return false;
  for (int h = 0; h < 2; ++h)
    if (b.w1_half[h].w && !(Cin == 2 * C && weight_is(b.w1_half[h], 27, C, C))) return false;
  return (b.w1_half[0].w == nullptr) == (b.w1_half[1].w == nullptr);
}

// the whole descriptor before anything is enqueued: the recursion of ublock() trusts `deeper`, the level count and every width
inline bool net_ok(const tl_net_desc* net) {
  const int nl = net->num_levels;
  if (!weight_is(net->w_in, 27, net->in_channels, net->u[0].C)) return false;
  if (net->u[0].C != 8 && net->u[0].C != 16 && net->u[0].C != 32 && net->u[0].C != 64) return false;        // the widths tl_head_mlp is instantiated for
  if (!affine_ok(net->out_bn) || !net->head_w1 || !net->head_b1 || !net->head_w2 || !net->head_b2) return false;
  for (int l = 0; l < nl; ++l) {
    const tl_ublock_desc& u = net->u[l];
    if (u.C <= 0 || (u.deeper != 0) != (l + 1 < nl)) return false;
    if (!res_ok(u.blocks[0], u.C, u.C) || !res_ok(u.blocks[1], u.C, u.C)) return false;
    if (!u.deeper) continue;
    const int Cd = net->u[l + 1].C;
    if (!affine_ok(u.bn_down) || !affine_ok(u.bn_up) || !affine_ok(u.bn_cat_l) || !affine_ok(u.bn_cat_r)) return false;
    if (!weight_is(u.wd, 8, u.C, Cd) || !weight_is(u.wu, 8, Cd, u.C)) return false;
    if (!res_ok(u.tail[0], 2 * u.C, u.C) || !res_ok(u.tail[1], u.C, u.C)) return false;
  }
  return true;
}

}  // namespace

extern "C" {

tl_exec* tl_exec_create(void) {
  tl_exec* ex = new (std::nothrow) tl_exec();
  if (!ex) return nullptr;
  if (hipHostMalloc(reinterpret_cast<void**>(&ex->host), kHostBytes, hipHostMallocDefault) != hipSuccess || (memset(ex->host, 0, kHostBytes), false) ||
      hipEventCreateWithFlags(&ex->ev_main, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ex->ev_side, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ex->ev_flag, hipEventDisableTiming) != hipSuccess) {
    tl_exec_destroy(ex);
    return nullptr;
  }
  return ex;
}

void tl_exec_destroy(tl_exec* ex) {
  if (!ex) return;
  for (hipEvent_t e : ex->pev) (void)hipEventDestroy(e);
  if (ex->ev_main) (void)hipEventDestroy(ex->ev_main);
  if (ex->ev_side) (void)hipEventDestroy(ex->ev_side);
  if (ex->ev_flag) (void)hipEventDestroy(ex->ev_flag);
  if (ex->host) (void)hipHostFree(ex->host);
  delete ex;
}

int tl_exec_check(tl_exec* ex) {
  if (!ex) return TL_ERR_ARG;
  if (ex->flag_in_flight) {
    if (hipEventSynchronize(ex->ev_flag) != hipSuccess) return TL_ERR_LAUNCH;
    ex->flag_in_flight = false;
  }
  if (ex->host[8]) {
    ex->host[8] = 0;
    return TL_ERR_BLK;
  }
  return TL_OK;
}

int tl_exec_profile(tl_exec* ex, int enable) {
  if (!ex) return TL_ERR_ARG;
  ex->profile = enable != 0;
  return TL_OK;
}

int tl_exec_profile_read(tl_exec* ex, tl_launch_rec* recs, int cap) {
  if (!ex || !recs || cap < 0) return TL_ERR_ARG;
  const int n = (int)ex->recs.size();
  if (n == 0) return 0;
  if (hipEventSynchronize(ex->pev.back()) != hipSuccess) return TL_ERR_LAUNCH;
  for (int i = 0; i < n && i < cap; ++i) {
    recs[i] = ex->recs[i];
    if (hipEventElapsedTime(&recs[i].ms, ex->pev[2 * i], ex->pev[2 * i + 1]) != hipSuccess) return TL_ERR_LAUNCH;
  }
  return n < cap ? n : cap;
}

int tl_forward(tl_exec* ex, const tl_net_desc* net, tl_forward_args* a, tl_stream_t stream) {
  if (!ex || !net || !a || !a->xyz || !a->batch_ids || !a->logits || !a->offsets || a->N <= 0 || a->B <= 0) return TL_ERR_ARG;
  if (net->num_levels < 2 || net->num_levels > TL_MAX_LEVELS || net->in_channels <= 0 || net->voxel_size <= 0.f) return TL_ERR_ARG;
  if (net->dtype != TL_F32 && net->dtype != TL_BF16 && net->dtype != TL_F16) return TL_ERR_ARG;
  if (!net_ok(net)) return TL_ERR_ARG;
  const bool feats = net->use_coords || net->use_feats;            // voxel-mean input features (tree_learn.py:149-155) instead of ones
  if (feats && (!a->point_feats || net->in_channels <= 3 || net->in_channels > 8 || net->max_points_per_voxel <= 0)) return TL_ERR_ARG;
  if (!a->arena || ((uintptr_t)a->arena) % 256) return TL_ERR_ARG;
  hipStream_t s = tl_s(stream);
  const int nl = net->num_levels;
  const int64_t N = a->N;
  a->needed_bytes = 0; a->blocked_used = 0; a->launches = 0;
  for (hipEvent_t e : ex->pev) (void)hipEventDestroy(e);
  ex->pev.clear(); ex->recs.clear();

  Run R;
  R.ex = ex; R.net = net; R.a = a; R.s = s; R.nl = nl;
  R.esize = net->dtype == TL_F32 ? 4 : 2;
  R.ar.base = static_cast<char*>(a->arena); R.ar.cap = a->arena_bytes; R.ar.dry = false;

  // ---- phase A: per-point voxel coordinates + grid extent (geometry.build_geometry step 1; tree_learn.py:133-135)
  const bool one = a->B == 1;                         // one tile: the atomic-free form, per-workgroup extents folded here after the read-back
  const int64_t o_pc = R.ar.take(16 * N), o_maxc = R.ar.take(one ? 16 * TL_POINT_COORDS_MAX_PARTS : 16), o_mm = R.ar.take(one ? 24 * 256 : 24 * (int64_t)a->B);
  if (R.ar.peak > R.ar.cap) { a->needed_bytes = R.ar.peak * 3; return TL_ERR_ARENA; }        // (a first guess; the exact figure follows once the counts are known)
  int32_t* pcoords = reinterpret_cast<int32_t*>(R.ar.at(o_pc));
  int rc, n_parts = 0;
  if (one) rc = tl_voxel_point_coords_one(a->xyz, a->batch_ids, N, net->voxel_size, reinterpret_cast<uint32_t*>(R.ar.at(o_mm)), pcoords,
                                          reinterpret_cast<int32_t*>(R.ar.at(o_maxc)), &n_parts, stream);
  else rc = tl_voxel_point_coords(a->xyz, a->batch_ids, N, a->B, net->voxel_size, reinterpret_cast<uint32_t*>(R.ar.at(o_mm)), pcoords,
                                  reinterpret_cast<int32_t*>(R.ar.at(o_maxc)), stream);
  if (rc != TL_OK) return rc;
  if (one) {
    if (hipMemcpyAsync(ex->host + 16, R.ar.at(o_maxc), 16 * (size_t)n_parts, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return TL_ERR_LAUNCH;   // host sync #1
    for (int j = 0; j < 4; ++j) ex->host[j] = 0;
    for (int q = 0; q < n_parts; ++q)
      for (int j = 0; j < 4; ++j) ex->host[j] = ex->host[j] > ex->host[16 + 4 * q + j] ? ex->host[j] : ex->host[16 + 4 * q + j];
  } else if (hipMemcpyAsync(ex->host, R.ar.at(o_maxc), 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return TL_ERR_LAUNCH;   // host sync #1
  ex->flag_in_flight = false;           // (the flag copy of the previous forward on this context was enqueued on a stream this forward just drained ...
  if (ex->host[3]) return TL_ERR_EXTENT;
  if (ex->host[8]) {                    // ... so its verdict is in: the block-local builder's error flag of the PREVIOUS forward, unless tl_exec_check took it)
    ex->host[8] = 0;
    return TL_ERR_BLK;
  }
  const int32_t extent[3] = {ex->host[0] + 1, ex->host[1] + 1, ex->host[2] + 1};
  int32_t shape[TL_MAX_LEVELS][3];
  for (int j = 0; j < 3; ++j) {
    shape[0][j] = net->has_shape ? net->spatial_shape[j] : extent[j];                  // tree_learn.py:86-87,165
    if (extent[j] > shape[0][j]) return TL_ERR_EXTENT;
  }
  for (int l = 1; l < nl; ++l)
    for (int j = 0; j < 3; ++j) {
      shape[l][j] = shape[l - 1][j] / 2;
      if (shape[l][j] <= 0) return TL_ERR_REACH_ZERO;
    }

  // ---- phase B: occupancy bitmaps + popcount prefixes of every level, one block [bitmaps u64 tw | prefixes u32 tw | counts u32 8 | scan scratch]
  int64_t nw[TL_MAX_LEVELS], woff[TL_MAX_LEVELS + 1];
  {
    int32_t d[4] = {a->B, extent[0], extent[1], extent[2]};
    woff[0] = 0;
    for (int l = 0; l < nl; ++l) {
      for (int j = 0; j < 4; ++j) R.lv[l].dims[j] = d[j];
      nw[l] = tl_nwords(tl_dims(d));
      woff[l + 1] = woff[l] + nw[l];
      for (int j = 1; j < 4; ++j) d[j] = (d[j] + 1) / 2;
    }
  }
  const int64_t tw = woff[nl];
  const int64_t n_ws = tl_pyramid_ws_words(R.lv[0].dims, nl, nullptr);
  if (n_ws < 0) return TL_ERR_ARG;
  const int64_t o_pyr = R.ar.take(4 * (3 * tw + 8 + n_ws));
  if (R.ar.peak > R.ar.cap) { a->needed_bytes = R.ar.peak * 2; return TL_ERR_ARENA; }
  char* p0 = R.ar.at(o_pyr);
  rc = tl_pyramid_build(pcoords, N, R.lv[0].dims, shape[0], nl, reinterpret_cast<uint64_t*>(p0), reinterpret_cast<uint32_t*>(p0 + 8 * tw),
                        reinterpret_cast<uint32_t*>(p0 + 12 * tw), reinterpret_cast<uint32_t*>(p0 + 12 * tw + 32), stream);
  if (rc != TL_OK) return rc;
  if (hipMemcpyAsync(ex->host, p0 + 12 * tw, 4 * nl, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return TL_ERR_LAUNCH;   // host sync #2
  for (int l = 0; l < nl; ++l) {
    R.lv[l].n = a->level_n[l] = (int64_t)(uint32_t)ex->host[l];
    if (R.lv[l].n <= 0) return TL_ERR_REACH_ZERO;
  }

  // ---- phase C: the geometry block (word offsets, 64-word aligned -- the layout of geometry.build_geometry step 3)
  const int64_t n1 = R.lv[0].n;
  // fp32 rows take the block-local order only in the parity-fast mode: every 27-tap conv of level 1 then needs its split-bf16 weights
  const tl_ublock_desc& u0 = net->u[0];
  const bool x3_l1 = u0.blocks[0].w1.x3 && u0.blocks[0].w2.x3 && u0.blocks[1].w1.x3 && u0.blocks[1].w2.x3 && u0.tail[0].w1_half[0].x3 && u0.tail[0].w1_half[1].x3 &&
                     u0.tail[0].w2.x3 && u0.tail[1].w1.x3 && u0.tail[1].w2.x3;
  R.blocked = net->blocked && (net->dtype != TL_F32 || x3_l1) && u0.C == 32 && u0.deeper && u0.tail[0].w1_half[0].w && u0.tail[0].w1_half[1].w &&
              n1 >= kBlkMinRows && n1 <= kBlkMaxRows;
  a->blocked_used = R.blocked;
  int64_t cur = 0;
  auto take = [&](int64_t words) { const int64_t o = cur; cur += al64(words); return o; };
  R.o_v2p = take(2 * N);
  for (int l = 0; l < nl; ++l) {
    LevelG& v = R.lv[l];
    const bool skip = R.blocked && l == 0;               // a blocked level needs neither canonical coordinates nor the canonical table
    if (!skip) { v.coords = take(4 * v.n); v.nbr = take(27 * v.n); }
    if (!skip && v.n >= kCompactMinRows) v.ct = take(10 * v.n);
    if (l + 1 < nl) v.child = take(8 * R.lv[l + 1].n);
  }
  int64_t o_o2n = -1, o_perm = -1, o_cnew = -1, o_bws = -1;
  if (R.blocked) {
    const int64_t nbws = tl_blk_ws_words(R.lv[0].dims);
    o_o2n = take(n1); o_perm = take(n1); o_cnew = take(4 * n1); R.o_unit = take(4 * n1); R.o_counter = take(64); R.o_halo = take(32 * n1);
    R.o_lrb = take(9 * n1); R.o_pmask = take(n1); o_bws = take(nbws);
    if (feats) R.o_nn = take(27 * n1);            // the input conv of real features gathers: the plain table in the block-local order (tl_blk.nn)
  }
  // inverse tables (pre-set to -1 by one fill).  Level 1's inverse conv (64 -> 32) runs on the gather-once kernel in the 16-bit dtypes and in
  // bf16x3: that kernel reads the packed form, 4 B per row instead of the 32 B of the one-hot table (59 MB less to fill, write and read per tile)
  const bool packed0 = nl > 1 && net->u[0].C == 32 && net->u[1].C == 64 && (net->dtype != TL_F32 || net->u[0].wu.x3) && tl_conv_one_hot_direct_enabled(n1);
  const int64_t o_m1 = cur;
  for (int l = 0; l + 1 < nl; ++l) {
    if (l == 0 && packed0) R.lv[l].invp = take(R.lv[l].n);
    else R.lv[l].inv = take(8 * R.lv[l].n);
  }
  const int64_t o_m1_end = cur;
  R.g0 = R.ar.take(4 * cur);

  // the activations: a dry run of the network over a copy of the allocator gives the exact peak before anything is enqueued
  {
    Run D = R;
    D.ar.dry = true; D.ar.base = nullptr;
    D.network();
    a->needed_bytes = D.ar.peak;
    if (D.ar.peak > R.ar.cap) return TL_ERR_ARENA;
  }

  auto W = [&](int64_t word_off) { return word_off < 0 ? nullptr : reinterpret_cast<int32_t*>(R.ar.at(R.g0 + 4 * word_off)); };
  tl_level arr[TL_MAX_LEVELS];
  for (int l = 0; l < nl; ++l) {
    LevelG& v = R.lv[l];
    tl_level& t = arr[l];
    for (int j = 0; j < 4; ++j) t.dims[j] = v.dims[j];
    t.n = v.n;
    t.bitmap = reinterpret_cast<const uint64_t*>(p0 + 8 * woff[l]); t.prefix = reinterpret_cast<const uint32_t*>(p0 + 8 * tw + 4 * woff[l]);
    t.coords = W(v.coords); t.nbr = W(v.nbr); t.compact = W(v.ct); t.child = W(v.child); t.parent = nullptr; t.inv = W(v.inv); t.inv_packed = W(v.invp); t.o2n = nullptr;
  }
  hipStream_t side = s;
  // whatever happens after the fork below, `s` waits for the side stream before this call returns: the caller may reuse or free the arena
  // on `s` as soon as it has the return code, and the unit builder writes into it
  struct Join {
    tl_exec* ex; hipStream_t s; hipStream_t* side; bool done = false;
    int join() {
      if (done || *side == s) return TL_OK;
      done = true;
      return (hipEventRecord(ex->ev_side, *side) == hipSuccess && hipStreamWaitEvent(s, ex->ev_side, 0) == hipSuccess) ? TL_OK : TL_ERR_LAUNCH;
    }
    ~Join() { (void)join(); }
  } joiner{ex, s, &side};
  if (R.blocked) {
    tl_blk bk{};
    bk.o2n = W(o_o2n); bk.perm = W(o_perm); bk.coords_new = W(o_cnew); bk.unit = W(R.o_unit); bk.counter = W(R.o_counter); bk.halo = W(R.o_halo);
    bk.lrb = reinterpret_cast<uint32_t*>(W(R.o_lrb)); bk.pmask = W(R.o_pmask); bk.cap_units = n1; bk.halo_max = TL_BLK_HALO_MAX; bk.nn = W(R.o_nn);
    rc = tl_blk_build(arr[0].bitmap, arr[0].prefix, R.lv[0].dims, n1, &bk, reinterpret_cast<uint32_t*>(W(o_bws)), 1, stream);
    if (rc != TL_OK) return rc;
    arr[0].o2n = bk.o2n;
    // the unit builder (instruction-bound) beside the rulebook kernels of the other levels when the caller gives a side stream
    if (a->side_stream && tl_s(a->side_stream) != s) {
      side = tl_s(a->side_stream);
      if (hipEventRecord(ex->ev_main, s) != hipSuccess || hipStreamWaitEvent(side, ex->ev_main, 0) != hipSuccess) return TL_ERR_LAUNCH;
    }
    rc = tl_blk_build(arr[0].bitmap, arr[0].prefix, R.lv[0].dims, n1, &bk, reinterpret_cast<uint32_t*>(W(o_bws)), 2, reinterpret_cast<tl_stream_t>(side));
    if (rc != TL_OK) return rc;
  }
  rc = tl_rulebooks_build(arr, nl, W(o_m1), o_m1_end - o_m1, pcoords, N, reinterpret_cast<int64_t*>(W(R.o_v2p)), stream);
  if (rc != TL_OK) return rc;
  if (joiner.join() != TL_OK) return TL_ERR_LAUNCH;
  // tl_blk_build's error flag (units skipped: cannot happen with cap_units >= n and halo_max <= 126, so this is an assertion) goes home behind
  // the builder, ahead of the convs: tl_exec_check waits for exactly this copy; a following forward of the context sees it after its first
  // read-back -- neither costs this forward a synchronisation
  if (R.blocked) {
    if (hipMemcpyAsync(ex->host + 8, W(R.o_counter) + 1, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipEventRecord(ex->ev_flag, s) != hipSuccess) return TL_ERR_LAUNCH;
    ex->flag_in_flight = true;
  }

  // ---- the network
  R.network();
  a->launches = R.launches;
  return R.rc;
}

}  // extern "C"
