/*
 * treelearn_hip.h -- C ABI of libtreelearn_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary of the TreeLearn per-tile sparse-conv segmentation path.  The
 * reference (ecker-lab/TreeLearn) has no FFI of its own: its device work is delegated to
 * the third-party `spconv` wheel and to ATen.  Each entry point below replaces one of
 * those call sites (cited as reference file:line).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (the PyTorch-ROCm caching
 *     allocator in our host code); the library never allocates or frees device memory;
 *   - every call takes the hipStream_t to enqueue on (as void*) and returns immediately;
 *   - return value: 0 = TL_OK, negative = error (tl_error_string); no exceptions; the only
 *     process-wide state is the set of developer switches behind tl_set_tuning (kernel-family
 *     selection for A/B runs and tests; results do not depend on them beyond re-association);
 *     calls on one stream must not be issued concurrently;
 *   - "table" = rulebook in tap-major layout i32[K][n_out], entry = input row or -1
 *     (SURVEY.md Appendix F canonical form, transposed for coalesced access).
 */
#ifndef TREELEARN_HIP_H
#define TREELEARN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tl_stream_t; /* hipStream_t */

enum {
  TL_OK = 0,
  TL_ERR_ARG = -1,       /* bad argument (null pointer, unsupported size) */
  TL_ERR_LAUNCH = -2,    /* hipGetLastError() after a launch was not hipSuccess */
  TL_ERR_UNSUPPORTED = -3
};

/* TL_F16 (IEEE half, BASELINE config 5's "fp16"; the reference trains under fp16 autocast + GradScaler, tools/training/train.py:32,40-44):
 * served by every entry point that takes a dtype -- the conv / head / weight-packing path and, since round 5, the training entry points
 * (tl_conv_wgrad*, tl_bn_train_*, the epilogue modes of tl_conv_fwd, tl_gather_rows / tl_scatter_add_rows, tl_linear_small_f32): the units are
 * compiled a second time for _Float16 (csrc/tl_half.h, tl_f16_train.h).  One 16-bit type per call (no bf16 / f16 mixtures).  Values beyond
 * 65504 become inf (no saturation), as on any fp16 path. */
enum { TL_F32 = 0, TL_BF16 = 1, TL_F16 = 2 };

int tl_version(void);
/* Developer tuning knobs, process-wide ("win" / "direct" / "stream" / "streamq" / "blk" / "up" / "direct_oh": enable a conv kernel
 * family or form -- "blk" the staged-unit kernel of the block-local level, "up" the scatter form of the inverse conv, "direct_oh" the
 * gather-once form of the level-1 inverse conv; "win_rows": window rows of the window kernel; "win_min_rows" / "small_rows": row
 * thresholds; "small_mode": variant of the small-level kernel; "bf16_depth": register prefetch depth 1..4).  Not needed for correct
 * results; the parity tests use them to force every family over the same data.  They are read on the host when a launch is
 * dispatched (launches already enqueued are unaffected); do not change them while another host thread is inside the library. */
int tl_set_tuning(const char* key, int64_t value);
const char* tl_error_string(int code);

/* ------------------------------------------------------------------ voxel hashing
 * Replaces spconv PointToVoxel.generate_voxel_with_id + the min/max/.tolist() prologue of
 * `voxelize` (reference tree_learn/model/tree_learn.py:129-167, call site :136-143).
 *
 * The "hash table" is a succinct rank structure instead of an open-addressing table:
 * an occupancy bitmap over the tile's voxel grid (1 bit per cell, z along the 64-bit word)
 * plus an exclusive prefix sum of word popcounts.  rank(cell) = prefix[word] +
 * popc(word & below(bit)) is the voxel's row in ascending (b,x,y,z) order, i.e. the
 * canonical order of SURVEY.md Appendix F, with no sort and no probing.
 */

/* Per-point integer voxel coordinates.
 *   xyz f32[N,3], batch_ids i64[N], B batch elements, voxel_size.
 *   ws_minmax u32[B*6] scratch; pcoords i32[N,4] out = (b,x,y,z) with
 *   x = floorf((p - min_b) / voxel_size) in fp32 (tree_learn.py:134, spconv PointToVoxel);
 *   maxc i32[4] out = {max x, max y, max z, error flag}. */
int tl_voxel_point_coords(const float* xyz, const int64_t* batch_ids, int64_t N, int B, float voxel_size,
                          uint32_t* ws_minmax, int32_t* pcoords, int32_t* maxc, tl_stream_t stream);

/* Occupancy bitmap of the level-1 grid.  dims = {B, X, Y, Z}; bitmap u64[B*X*Y*ceil(Z/64)] (zeroed here). */
/* The same for ONE tile (B = 1, what the tile loop's forward passes: tree_learn/util/pipeline.py:83-88), without same-address atomics: the
 * per-workgroup coordinate maxima are written as rows of `maxc_parts` i32[TL_POINT_COORDS_MAX_PARTS][4] = (max x, max y, max z, error
 * flag); *n_parts (host) rows are valid and the caller folds them (max / or) after its read-back.  ws: u32[6 * 256].  pcoords as above. */
#define TL_POINT_COORDS_MAX_PARTS 1024
int tl_voxel_point_coords_one(const float* xyz, const int64_t* batch_ids, int64_t N, float voxel_size, uint32_t* ws, int32_t* pcoords,
                              int32_t* maxc_parts, int* n_parts, tl_stream_t stream);
int tl_bitmap_from_points(const int32_t* pcoords, int64_t N, const int32_t dims[4], uint64_t* bitmap, tl_stream_t stream);

/* Occupancy of the stride-2 coarse grid (SparseConv3d k=2 s=2 output set, reference
 * tree_learn/model/blocks.py:104-110): coarse cell = OR of its 2x2x2 children; cells at or beyond
 * out_shape[3] (= in_shape // 2) are dropped.
 * fdims = {B,Xf,Yf,Zf}; cdims = {B, ceil(Xf/2), ceil(Yf/2), ceil(Zf/2)}. */
int tl_bitmap_down(const uint64_t* fine, const int32_t fdims[4], const int32_t out_shape[3],
                   uint64_t* coarse, const int32_t cdims[4], tl_stream_t stream);

/* Exclusive prefix sum of popcounts.  prefix u32[nwords]; total u32[1] (device); ws u32[tl_scan_ws_words(nwords)]. */
int64_t tl_scan_ws_words(int64_t nwords);
int tl_bitmap_scan(const uint64_t* bitmap, int64_t nwords, uint32_t* prefix, uint32_t* total, uint32_t* ws, tl_stream_t stream);

/* coords i32[M,4] = (b,x,y,z) of every set bit, in rank order. */
int tl_expand_coords(const uint64_t* bitmap, const uint32_t* prefix, const int32_t dims[4], int32_t* coords, tl_stream_t stream);

/* v2p i64[N]: voxel row of every point (tree_learn.py:143 pc_voxel_id, :160-161 batch offsets implicit). */
int tl_point_rank(const int32_t* pcoords, int64_t N, const uint64_t* bitmap, const uint32_t* prefix,
                  const int32_t dims[4], int64_t* v2p, tl_stream_t stream);

/* Optional voxel features (use_coords / use_feats, tree_learn.py:149-156): mean over the first <= P points of
 * each voxel in input order, all-zero point rows skipped; pf f32[N,C] rows (x,y,z,feat..); out f32[M,C]
 * in (x,y,z,feat..) order; ws i32[M*P] scratch. */
int tl_voxel_mean_feats(const float* pf, int C, const int64_t* v2p, int64_t N, int64_t M, int P,
                        int32_t* ws, float* out, tl_stream_t stream);
/* The same means straight into the input conv's operand (tree_learn.py:149-156 in one launch sequence): columns (x, y, z) from
 * xyz f32[N,3] and f_0..f_{F-1} from feats f32[N,F] (F <= 5), column groups switched off by use_coords / use_feats = 0 set to 1,
 * output rows in the reference's (feat.., x, y, z) order, cast to `dtype` (round to nearest even): out [M, 3 + F]; ws i32[M*P]. */
int tl_voxel_feats(const float* xyz, const float* feats, int F, const int64_t* v2p, int64_t N, int64_t M, int P, int use_coords,
                   int use_feats, int dtype, int32_t* ws, void* out, tl_stream_t stream);

/* ------------------------------------------------------------------ rulebooks
 * Replaces spconv's indice generation for SubMConv3d(k=3,pad=1) (`subm{l}`, tree_learn.py:37-39,
 * blocks.py:57-70) and SparseConv3d(k=2,s=2)/SparseInverseConv3d(k=2) (`spconv{l}`, blocks.py:104-123). */

/* nbr i32[27][M]: tap = (dx+1)*9 + (dy+1)*3 + (dz+1), entry = input row or -1; nbr[13][i] == i. */
int tl_rulebook_subm(const int32_t* coords, int64_t M, const uint64_t* bitmap, const uint32_t* prefix,
                     const int32_t dims[4], int32_t* nbr, int32_t* compact /* optional i32[10][M] column form, see tl_rulebook_compact */,
                     tl_stream_t stream);

/* child i32[8][Mc] (tap = (x&1)*4+(y&1)*2+(z&1)), parent i32[Mf] (-1 = dropped), inv i32[8][Mf]
 * (inv[tap(p)][p] = parent[p], other taps -1: the inverse conv's table).  parent and inv are filled here. */
int tl_rulebook_down(const int32_t* ccoords, int64_t Mc, const uint64_t* fbitmap, const uint32_t* fprefix,
                     const int32_t fdims[4], int64_t Mf, int32_t* child, int32_t* parent, int32_t* inv, tl_stream_t stream);

/* ---- the whole pyramid in two calls (what treelearn_amd.geometry uses; same kernels and results as the per-level calls
 * above, enqueued back to back, the deep levels -- a few thousand words each -- batched into shared launches).
 *
 * tl_pyramid_ws_words: scratch (u32 words: the scans' partial sums + a byte map of level 1, 64 B per bitmap word) for a pyramid whose level 1
 *   has dims0 = {B,X,Y,Z}; level l+1 has
 *   ceil(dims_l / 2).  level_word_offsets (optional, i64[num_levels+1]) receives the word offset of every level inside the
 *   `bitmaps` / `prefixes` arrays (the last entry = total words).
 * tl_pyramid_build: occupancy bitmap of level 1 from pcoords, then per level the k2s2 down-sampling (cells at or beyond
 *   shape_l = shape0 >> l dropped, spconv's floor((n-2)/2)+1 output shape; tree_learn.py:86-87, blocks.py:104-108) and the
 *   popcount scan; counts u32[num_levels] (device) = active voxels per level. */
int64_t tl_pyramid_ws_words(const int32_t dims0[4], int num_levels, int64_t* level_word_offsets);
int tl_pyramid_build(const int32_t* pcoords, int64_t N, const int32_t dims0[4], const int32_t shape0[3], int num_levels,
                     uint64_t* bitmaps, uint32_t* prefixes, uint32_t* counts, uint32_t* ws, tl_stream_t stream);

/* One level of the pyramid for tl_rulebooks_build: inputs dims/n/bitmap/prefix, outputs as in tl_expand_coords,
 * tl_rulebook_subm (nbr, optional compact) and tl_rulebook_down (child/parent/inv: the tables between this level and the
 * next coarser one; unused on the last level). */
typedef struct tl_level {
  int32_t dims[4];
  int64_t n;
  const uint64_t* bitmap;
  const uint32_t* prefix;
  int32_t* coords;
  int32_t* nbr;
  int32_t* compact;
  int32_t* child;
  int32_t* parent;
  int32_t* inv;
  /* optional (NULL = canonical order): the block-local order of THIS level (tl_blk_build's o2n).  The tables that hold or are indexed
   * by rows of this level -- child (entries), parent / inv (index) and, on level 1, v2p (entries) -- are then written in that order.
   * With o2n set, coords / nbr / compact of the level may be NULL (the block-local conv kernel needs none of them). */
  const int32_t* o2n;
  /* optional: the inverse table of this level in its packed form i32[n] = (parent row << 3) | tap, -1 = no parent -- 4 B per row instead of the
   * 32 B of the one-hot `inv` (tl_conv_args.table_one_hot = 2 reads it).  With inv_packed set, `inv` may be NULL; `parent` may always be NULL. */
  int32_t* inv_packed;
} tl_level;

/* coords + every rulebook of every level + (v2p != NULL) the point -> voxel map, in one call.  parent / inv arrays that lie
 * inside [minus_one, minus_one + minus_one_words) are set to -1 by a single fill of that block (optional; others are filled
 * one by one). */
int tl_rulebooks_build(const tl_level* levels, int num_levels, int32_t* minus_one, int64_t minus_one_words,
                       const int32_t* pcoords, int64_t N, int64_t* v2p, tl_stream_t stream);

/* ---- block-local row order + staged rulebook of a level (what the level-1 convs of the inference engine run on; csrc/tl_blk.hip).
 * The reference's spconv keeps voxels in hash order and gathers every tap of every output row (blocks.py:57-70 -> spconv's
 * implicit-GEMM gather); the canonical order here (ascending key) pins the bit-exact rulebook tests.  For the matrix-core kernel a
 * THIRD, internal order is better: voxels sorted by 8x8x8 block -- blocks ordered by (batch, x >> 5, y >> 5, (x >> 3) & 3, (y >> 3) & 3,
 * z >> 3), i.e. tiles of 4 x 4 block columns --, canonical order inside a block.  Units of <= 64
 * consecutive rows of that order then reach few rows outside themselves (their "halo"), so a wave can stage own + halo rows once in
 * LDS and read all 27 taps from there.
 *   o2n / perm     i32[n]: canonical row -> new row and back;  coords_new i32[n][4] = (b,x,y,z) in the new order
 *   unit           i32[cap_units][4] = {first new row, rows (1..64), halo rows, 0}.  Chunk c = new rows [64 c, 64 c + 64) is unit c when
 *                  its halo has <= halo_max rows; otherwise it is halved (recursively) and the pieces after the first are appended
 *                  behind the ceil(n / 64) regular units, ascending by first row.  counter[0] (DEVICE) = number of units, counter[1] = error flag
 *   halo           i32[32 n]: the unit that starts at row r0 lists its DISTINCT outside rows ascending at halo + 32 r0, padded with -1
 *                  to a multiple of 16
 *   lrb            u32[n][9]: 27 ten-bit entries per row, tap k in word k / 3 at bits 10 (k % 3): entry = 4 * pos + ((pos >> 2) & 3), so that
 *                  entry * 16 is the byte offset of tap k's input row in the unit's stage (64-B rows, 16-B pieces XOR-swizzled), with
 *                  pos = own row index (0..63), 64 + halo rank, or 191 = absent (the stage's zero row)
 *   pmask          i32[n]: 27-bit presence mask of the row's taps
 * n < 2^25.  ws u32[tl_blk_ws_words(dims)] (16-B aligned).  Deterministic: the appended units are sorted by their first row. */
#define TL_BLK_HALO_MAX 126
typedef struct tl_blk {
  int32_t* o2n; int32_t* perm; int32_t* coords_new;
  int32_t* unit; int32_t* counter; int32_t* halo; uint32_t* lrb; int32_t* pmask;
  int64_t cap_units;        /* rows of `unit` (>= ceil(n / 64); n always suffices) */
  int32_t halo_max;         /* 26 .. TL_BLK_HALO_MAX */
  int32_t reserved;
  int32_t* nn;              /* optional (NULL = not wanted): the rulebook as a plain table i32[27][n] in the NEW row order (entry = new input row or
                             * -1) -- what the kernels without a staged form read (weight gradient, convs of other widths) */
} tl_blk;
int64_t tl_blk_ws_words(const int32_t dims[4]);
/* phases: 1 = the order (o2n, perm, coords_new + the scratch the second phase reads), 2 = units / halo / lrb / pmask, 3 = both.  The
 * two phases may be enqueued on different streams (phase 2 after phase 1): what tl_rulebooks_build needs of a blocked level is o2n only,
 * so the unit builder -- instruction-bound, light on memory -- can run beside the rulebook kernels of the other levels. */
int tl_blk_build(const uint64_t* bitmap, const uint32_t* prefix, const int32_t dims[4], int64_t n, const tl_blk* out, uint32_t* ws,
                 int phases, tl_stream_t stream);

/* Column form of a 27-tap SubM rulebook built by tl_rulebook_subm: compact i32[10][n] = the row of the first present
 * dz neighbour of each of the 9 (dx, dy) columns (or -1) followed by a 27-bit presence mask.  Present neighbours of a
 * column are consecutive rows, so entry k = 3 c + d is compact[c] + popcount(mask bits 3c .. k-1).  tl_conv_fwd reads this
 * (40 B per voxel) instead of the table (108 B) in the kernels that support it (tl_conv_args.table_compact). */
int tl_rulebook_compact(const int32_t* table, int64_t n, int32_t* compact, tl_stream_t stream);

/* ------------------------------------------------------------------ sparse convolution
 * Replaces spconv SubMConv3d / SparseConv3d / SparseInverseConv3d forward (blocks.py:57-70,104-123,
 * tree_learn.py:37-39) and Custom1x1Subm3d's torch.mm (blocks.py:29-39), with the surrounding
 * BatchNorm1d+ReLU (blocks.py:56,63,103,117; tree_learn.py:42), residual add (blocks.py:76) and skip
 * concat (blocks.py:146, via in_ld/out_ld column views) fused as prologue / epilogue.
 *
 *   out[o, :] = epi( sum_k  W[k] . pro(in[table[k][o], :]) )        (absent rows contribute 0)
 *   pro(x) = relu?(x * in_scale + in_shift)       epi(y) = relu?((y + residual[o]) * out_scale + out_shift)
 *   out2 / out3 (optional) = relu?((y + residual[o]) * out{2,3}_scale + out{2,3}_shift)
 */
typedef struct tl_conv_args {
  const void* in;        int64_t in_ld;   /* row stride in elements (>= Cin) */
  const void* weight;                     /* [K][Cout][Cin], same dtype as `in` */
  const int32_t* table;                   /* [K][n_out], or NULL = identity (K must be 1) */
  int64_t n_out;         int64_t n_in;
  int32_t K;             int32_t Cin;     int32_t Cout;   int32_t dtype;   /* TL_F32 | TL_BF16 | TL_F16 */
  const float* in_scale; const float* in_shift;  int32_t in_relu;  int32_t out_relu;
  const void* residual;  int64_t res_ld;
  const float* out_scale; const float* out_shift;
  void* out;             int64_t out_ld;
  /* up to two extra views of y = acc + residual, each with its own affine/ReLU -- lets a producer store the
   * raw tensor (for the residual branch) AND relu(bn_next(y)) (for the next conv's gathers) in one pass: */
  void* out2; int64_t out2_ld; const float* out2_scale; const float* out2_shift; int32_t out2_relu;
  void* out3; int64_t out3_ld; const float* out3_scale; const float* out3_shift; int32_t out3_relu;
  /* optional second copy of the weights in MFMA-fragment order (tl_pack_weight_frag), NULL if absent: kernels that
   * read B operands straight from global memory (the small-level kernel) then load 1 KB contiguous per instruction */
  const void* weight_frag;
  /* != 0: every output row has at most ONE valid table entry (SparseInverseConv3d: the row's parent through its own
   * octant tap).  Kernels may then gather that single row once and route it to its tap instead of issuing K gathers.
   * 2: `table` is the PACKED form i32[n_out] = (input row << 3) | tap, -1 = none (tl_level.inv_packed; K = 8) -- served by the gather-once
   * kernels only (16-bit and bf16x3, 64 -> 32): any other shape returns TL_ERR_UNSUPPORTED rather than misreading the table. */
  int32_t table_one_hot;
  /* optional column form of `table` (tl_rulebook_compact; K must be 27), NULL if absent */
  const int32_t* table_compact;
  /* Training-mode epilogue reductions (the BatchNorm1d that surrounds every conv of the reference, blocks.py:55-70,102-123, in
   * train() mode; tools/training/train.py:30-44).  With y = acc + residual:
   *   TL_EPI_STATS : red_part f64[parts][2][Cout] receives per-workgroup sums of y and y^2 (y as stored) -- the batch statistics of
   *                  the BatchNorm that consumes this conv's output (finish: tl_bn_train_finish), without a pass over y;
   *   TL_EPI_BN_BWD: this launch is the input-gradient conv of a layer whose input was relu?(bn(x)): y is dy of that activation;
   *                  every view receives g = dy * [x * bn_scale + bn_shift > 0] (bn_relu) and red_part the sums of g and of
   *                  g * (x - bn_mean) * bn_rstd = dbeta / dgamma (finish + dx: tl_bn_train_bwd_from_parts).  bn_x [n_out, Cout]
   *                  in `dtype`, row stride bn_x_ld.
   * red_part must hold tl_conv_red_parts(n_out) rows; *red_nparts (HOST, optional) receives the rows actually written.
   * Only the direct / stream kernel families carry these epilogues: other shapes return TL_ERR_UNSUPPORTED (nothing launched)
   * and the caller runs the separate passes (tl_bn_train_stats / tl_bn_train_bwd).  Deterministic. */
  /* != 0: every element of `in` is 1 (the reference's default use_feats = False, use_coords = False feeds all-ones voxel features,
   * tree_learn.py:129-167): out[o] = sum over the present taps of sum_c W[k][:][c].  Served without reading `in` when K = 27,
   * Cout = 32, bf16 and table_compact is given; otherwise the flag is ignored (the general kernels read the ones). */
  int32_t in_all_ones;
  int32_t epi_mode;
  double* red_part;
  int32_t* red_nparts;
  const void* bn_x;      int64_t bn_x_ld;
  const float* bn_mean;  const float* bn_rstd;  const float* bn_scale;  const float* bn_shift;  int32_t bn_relu;
  /* Block-local form of a 27-tap SubM rulebook (tl_blk_build; all five NULL when absent).  `in`, `out*`, `residual` are then in the
   * block-local row order, `table` may be NULL, and K = 27, Cin = Cout = 32, TL_BF16 / TL_F16 are served by the staged-unit kernel
   * (csrc/tl_conv_blk.hip; same summation order as the gather kernels: bit-identical results): up to three views; the gather-side
   * prologue (in_scale / in_shift / in_relu) applied once per STAGED row, with one view; the training epilogues (TL_BF16, one view).
   * With in_all_ones the presence masks come from blk_pmask.  Other shapes fall through to `table` (TL_ERR_UNSUPPORTED without one). */
  const int32_t* blk_unit; const int32_t* blk_counter; const int32_t* blk_halo; const uint32_t* blk_lrb; const int32_t* blk_pmask;
  /* Scatter form of a one-hot table (table_one_hot = 1, K = 8: the inverse conv of reference blocks.py:113-123), i32[K][n_in]: the output
   * row that input row c feeds through tap k, or -1 -- i.e. the rulebook of the stride-2 conv whose pairs the inverse conv re-uses
   * (tl_level.child).  Optional (NULL: the gather forms run); when given, 16-bit launches of the widths of levels 1-4 walk the INPUT rows
   * instead: each read once, eight small products, rows scattered to the children that exist (csrc/tl_conv_up.hip).  Same results. */
  const int32_t* table_scatter;
  /* Optional split-bf16 copy of fp32 weights (tl_pack_weight_x3; TL_F32 launches only, NULL = absent).  When given, the kernel families
   * that serve the large levels (csrc/tl_conv_direct.hip, tl_conv_stream.hip) contract in the "bf16x3" form: every fp32 operand x is split in
   * registers into hi = bf16(x) and lo = bf16(x - hi), and a . b is formed as alo.bhi + ahi.blo + ahi.bhi on the bf16 matrix cores with fp32
   * accumulation -- storage, BatchNorm, residual and epilogues stay fp32, the contraction costs 3/16 of the fp32-input MFMA, products are
   * exact to ~2^-15 relative (the reference's fp32 inference, tree_learn/util/pipeline.py:86, within the 1e-3 parity gate at a third of
   * the exact mode's time).  Shapes without such an instantiation run the exact fp32 kernels on `weight`. */
  const void* weight_x3;
} tl_conv_args;

#define TL_EPI_NONE 0
#define TL_EPI_STATS 1
#define TL_EPI_BN_BWD 2
/* upper bound of the partial rows a tl_conv_fwd launch with epi_mode != 0 writes for n_out output rows */
int64_t tl_conv_red_parts(int64_t n_out);

int tl_conv_fwd(const tl_conv_args* args, tl_stream_t stream);

/* Repack a reference-layout conv weight [Cout, k,k,k, Cin] (spconv `.weight`)
 * into the kernel layout [K=k^3][Cout][Cin] with dtype conversion. */
int tl_pack_weight(const float* w_ref, int Cout, int K, int Cin, void* w_packed, int dtype, tl_stream_t stream);
/* The split-bf16 form of a reference-layout fp32 conv weight for tl_conv_args.weight_x3 (Cin % 32 == 0): [K][Cout][Cin / 32][64 bf16] =
 * per 32-channel unit the hi parts bf16(w) of its four 8-channel MFMA pieces, then the lo parts bf16(w - hi): as many bytes as [K][Cout][Cin] fp32;
 * for Cin < 256 and Cout % 32 == 0 a second copy of the same bytes in MFMA-fragment order follows (1 KB contiguous per fragment load: the
 * small-level kernel).  tl_pack_weight_x3_bytes = the size of the buffer to pass. */
int64_t tl_pack_weight_x3_bytes(int Cout, int K, int Cin);
int tl_pack_weight_x3(const float* w_ref, int Cout, int K, int Cin, void* w_x3, tl_stream_t stream);
/* Weights of the input-gradient ("dgrad") conv of the same layer, for tl_conv_fwd over the transposed rulebook: w_t[k][ci][co] =
 * w_ref[co][flip ? K-1-k : k][ci] (taps flip for SubM convs: nbr[k][o] = i <=> nbr[K-1-k][i] = o), with dtype conversion. */
int tl_pack_weight_dgrad(const float* w_ref, int Cout, int K, int Cin, int flip, void* w_t, int dtype, tl_stream_t stream);
/* All three packings for many layers in ONE launch (a training step re-packs every conv weight after the optimizer: ~210 launches per
 * step otherwise).  descs (DEVICE array): src = the fp32 parameter [Cout][K][Cin], dst = the packed tensor in `dtype`, form 0 = tl_pack_weight,
 * 1 = tl_pack_weight_frag, 2 / 3 = tl_pack_weight_dgrad without / with flipped taps.  blocks (DEVICE, i32[n_blocks][2]): workgroup b writes
 * the 4096 output elements of descriptor blocks[b][0] starting at element 4096 * blocks[b][1]. */
typedef struct tl_pack_desc {
  const float* src;
  void* dst;
  int32_t Cout, K, Cin, form;
} tl_pack_desc;
int tl_pack_weights_batch(const tl_pack_desc* descs, const int32_t* blocks, int64_t n_blocks, int dtype, tl_stream_t stream);
/* Same weights in fragment order (Cout % 32 == 0, Cin % 32 == 0): [K][Cout/32][Cin/32][J][64 lanes][16 B], where lane
 * (fi = lane & 31, fh = lane >> 5) of 16-B piece j holds W[k][32 cb + fi][32 ch + (32 j + 16 fh) / sizeof(elem) ...]:
 * the B operand of one 32x32 MFMA step is one contiguous 1 KB block (J = 2 for bf16, 4 for fp32). */
int tl_pack_weight_frag(const float* w_ref, int Cout, int K, int Cin, void* w_frag, int dtype, tl_stream_t stream);

/* Weight gradient (training step; spconv's autograd behind the conv modules, tools/training/train.py:40):
 *   gw[k][co][ci] = sum_o gout[o][co] * x[table[k][o]][ci]   over the present rulebook entries, fp32.
 * x [n_in, x_ld >= Cin] and gout [n_out, g_ld >= Cout] in `dtype` (TL_F32, or TL_BF16 = mixed-precision training: widened
 * to fp32 in registers; TL_F16 likewise), table i32[K][n_out] or NULL (K = 1: identity), gw f32[K][Cout][Cin] out (fully written),
 * ws f32[tl_conv_wgrad_ws_floats(...)] scratch.  Deterministic. */
int64_t tl_conv_wgrad_ws_floats(int64_t n_out, int K, int Cin, int Cout);
int tl_conv_wgrad(const void* x, int64_t x_ld, const void* gout, int64_t g_ld, int dtype, const int32_t* table, int64_t n_out,
                  int64_t n_in, int K, int Cin, int Cout, float* gw, float* ws, tl_stream_t stream);
/* the same with gw written in the reference parameter layout f32[Cout][K][Cin] (= spconv `.weight` [Cout,k,k,k,Cin]: the `.grad` of the
 * module parameter, no transposing copy afterwards); Cin % 4 == 0 */
int tl_conv_wgrad_ref(const void* x, int64_t x_ld, const void* gout, int64_t g_ld, int dtype, const int32_t* table, int64_t n_out,
                      int64_t n_in, int K, int Cin, int Cout, float* gw, float* ws, tl_stream_t stream);

/* The weight gradient of a 27-tap 32 -> 32 SubM conv over a BLOCK-LOCAL level (tl_blk_build: x and gout rows in the new order) in the
 * staged-unit form of the forward kernel: a unit's own and halo rows of x, its gout rows and its local rulebook are staged in LDS once and all
 * 27 taps contract against them, instead of 27 gathers per output row through the plain table (level 1 of the training step,
 * tools/training/train.py:40).  16-bit dtypes, Cin = Cout = 32; anything else returns TL_ERR_UNSUPPORTED and the caller uses tl_conv_wgrad over
 * tl_blk.nn.  gw f32[27][32][32], or f32[32][27][32] with ref_layout != 0; ws f32[tl_conv_wgrad_blk_ws_floats()].  Deterministic. */
int64_t tl_conv_wgrad_blk_ws_floats(void);
int tl_conv_wgrad_blk(const void* x, int64_t x_ld, const void* gout, int64_t g_ld, int dtype, const int32_t* blk_unit, const int32_t* blk_counter,
                      const int32_t* blk_halo, const uint32_t* blk_lrb, int64_t n, int Cin, int Cout, float* gw, int ref_layout, float* ws,
                      tl_stream_t stream);

/* ------------------------------------------------------------------ per-point heads
 * Replaces forward_head (tree_learn.py:97-103): features[v2p] gather, output_layer BN+ReLU
 * (tree_learn.py:42,93) as prologue, then both MLPs (blocks.py:8-18) with eval-mode BN folded:
 *   h = relu(W1' f + b1'),  y = W2 h + b2.
 * feats [M,C] (dtype), v2p i64[N]; w1 f32[2][C][C], b1 f32[2][C] (BN folded), w2 f32[5][C] (2 semantic rows
 * then 3 offset rows), b2 f32[5].  backbone f32[N,C] may be NULL (skip the 128 B/point write). */
int tl_head_mlp(const void* feats, int64_t feats_ld, int dtype, int C, const int64_t* v2p, int64_t N,
                const float* pro_scale, const float* pro_shift,
                const float* w1, const float* b1, const float* w2, const float* b2,
                float* backbone, float* logits, float* offsets, tl_stream_t stream);

/* ------------------------------------------------------------------ elementwise helpers */
/* out = relu?(in * scale + shift) over [n, C] with row strides (BatchNorm1d eval + ReLU, tree_learn.py:42). */
int tl_affine_relu(const void* in, int64_t in_ld, void* out, int64_t out_ld, int64_t n, int C, int dtype,
                   const float* scale, const float* shift, int relu, tl_stream_t stream);

/* ------------------------------------------------------------------ BatchNorm1d, training mode (+ ReLU)
 * The `norm_fn(C), nn.ReLU()` pair in front of every conv and inside both heads in TRAINING mode (reference
 * tree_learn/model/blocks.py:55-70,102-123, tree_learn.py:34-46: BatchNorm1d(eps=1e-4, momentum=0.1) over all active voxels of
 * the batch; exercised by tools/training/train.py:30-44).  Replaces ATen's batch_norm / relu forward and backward kernels.
 * Deterministic (fixed row partition, fp64 partial sums added in a fixed order, no atomics).
 *
 * tl_bn_train_stats: x [n, C] (f32 or bf16, row stride ld) -> mean, rstd = 1/sqrt(var_biased + eps), scale = gamma * rstd,
 *   shift = beta - mean * scale (all f32[C]); running_mean / running_var (nullable pair) get the momentum update with the
 *   UNBIASED variance like torch, num_batches_tracked (nullable, i64[1]) += 1.  y = relu(x * scale + shift) is then one
 *   tl_affine_relu call.  ws f64[tl_bn_ws_doubles(n, C)].  C % 4 == 0, C <= 1024.
 * tl_bn_train_bwd: dy [n, C] (f32 or bf16) = gradient w.r.t. y -> dx [n, C] in x's dtype, dgamma, dbeta f32[C]; `relu` != 0
 *   masks dy where y <= 0 (y is recomputed from x); dx_add (nullable, x's dtype, [n, C]) is added to dx in the same pass -- the
 *   gradient reaching x along its OTHER use (the residual / skip connection: x feeds both this BatchNorm and an identity path), so
 *   autograd's separate accumulation kernel disappears.  Under mixed precision (the reference trains under autocast,
 *   tools/training/train.py:32) activations and their gradients stay bf16 between layers; all statistics are fp64 / fp32. */
int64_t tl_bn_ws_doubles(int64_t n, int C);
int tl_bn_train_stats(const void* x, int64_t ld, int64_t n, int C, int dtype, const float* gamma, const float* beta, float eps,
                      float momentum, double* ws, float* mean, float* rstd, float* scale, float* shift, float* running_mean,
                      float* running_var, int64_t* num_batches_tracked, tl_stream_t stream);
int tl_bn_train_bwd(const void* x, int64_t ld, int x_dtype, const void* dy, int64_t dld, int dy_dtype, int64_t n, int C,
                    const float* mean, const float* rstd, const float* scale, const float* shift, int relu, double* ws,
                    float* dgamma, float* dbeta, void* dx, int64_t xld, const void* dx_add, int64_t dx_add_ld, tl_stream_t stream);

/* The same BatchNorm with its reductions taken from a conv epilogue (tl_conv_args.epi_mode): part f64[nparts][2][C] as written by
 * tl_conv_fwd.  tl_bn_train_finish = the second half of tl_bn_train_stats (partial sums of x and x^2 -> mean .. shift, running
 * statistics); tl_bn_train_bwd_from_parts = the second half of tl_bn_train_bwd for an ALREADY masked g (TL_EPI_BN_BWD: sums of g and
 * g * xhat -> dgamma, dbeta, then dx = scale * (g - dbeta / n - xhat * dgamma / n) + dx_add in one pass over x and g; C % 8 == 0 and
 * 16-B aligned rows, else TL_ERR_UNSUPPORTED). */
int tl_bn_train_finish(const double* part, int64_t nparts, int64_t n, int C, const float* gamma, const float* beta, float eps, float momentum, float* mean,
                       float* rstd, float* scale, float* shift, float* running_mean, float* running_var, int64_t* num_batches_tracked, tl_stream_t stream);
int tl_bn_train_bwd_from_parts(const void* x, int64_t ld, int x_dtype, const void* g, int64_t gld, int g_dtype, int64_t n, int C, const float* mean,
                               const float* rstd, const float* scale, const float* shift, const double* part, int64_t nparts, float* dgamma, float* dbeta,
                               void* dx, int64_t xld, const void* dx_add, int64_t dx_add_ld, tl_stream_t stream);

/* out f32[n, Cout] = x[n, Cin] . W^T for Cout <= 8 (the output Linears of the two heads in training, blocks.py:8-26 `MLP`): x and W
 * [Cout][Cin] in `dtype`, fp32 accumulation AND fp32 result -- under mixed precision the logits / offsets keep fp32 resolution. */
int tl_linear_small_f32(const void* x, int64_t x_ld, int dtype, const void* w, int Cin, int Cout, int64_t n, float* out, int64_t out_ld,
                        tl_stream_t stream);

/* Voxel -> point feature gather and its gradient (tree_learn.py:98 `output.features[v2p_map]`; tools/training/train.py:40):
 *   tl_gather_rows     : out[p, :] = in[idx[p], :] for p < N (idx < 0 counts from the end like torch indexing; in [n_rows, C]);
 *   tl_scatter_add_rows: gin[v, :] = sum over the points p with idx[p] == v of g[p, :], added in ascending p in fp32; `order` i64[N] =
 *                        the STABLE argsort of idx, sorted_idx = idx[order]; gin [n_rows, C] contiguous, fully written (rows without a
 *                        point are zero).  Deterministic, no atomics.  C * sizeof(elem) % 16 == 0, 16-B aligned rows. */
int tl_gather_rows(const void* in, int64_t in_ld, int dtype, int C, int64_t n_rows, const int64_t* idx, int64_t N, void* out, int64_t out_ld, tl_stream_t stream);
int tl_scatter_add_rows(const void* g, int64_t g_ld, int dtype, int C, const int64_t* order, const int64_t* sorted_idx, int64_t N, int64_t n_rows, void* gin,
                        int64_t gin_ld, tl_stream_t stream);

/* Keep rows where mask != 0 (masks_inner filtering before D2H, util/pipeline.py:100-103).
 * in f32[n,C] -> out f32[count,C]; count i32[1] device.  Stable (input order kept).
 * ws i32[tl_compact_ws_words(n)] scratch. */
int64_t tl_compact_ws_words(int64_t n);
int tl_compact_rows(const float* in, int C, const uint8_t* mask, int64_t n, float* out, int32_t* count,
                    int32_t* ws, tl_stream_t stream);

/* ------------------------------------------------------------------ inference tiling
 * Replaces, per tile, the box masks + `.cpu()` + np.savez / np.load round trip of
 * SampleGenerator.tile_generate_and_save (tree_learn/util/data_preparation.py:393-441,456-476) and the
 * bookkeeping of TreeDataset.__getitem__ (tree_learn/dataset/dataset.py:34-76,87-91), plot arrays resident in HBM.
 *   xyz f32[n,3], label f32[n], feat f32[n,F] : the voxelised plot;
 *   box (HOST struct): outer = (xmin,xmax,ymin,ymax) as float32 (closed box, compared in float32),
 *       inner = the inner square in float64 (x in [x0,x1), y in (y0,y1], compared in float64),
 *       center = tile centre subtracted in float64, half_inner = inner_square_edge_length / 2;
 *   outputs, capacity n rows, rows kept in plot order: coords f32[.,3] (centred), out_feat f32[.,F],
 *   instance_labels i64 (int32 truncation of label), semantic_labels i64 (label 0 -> 1, else 0),
 *   mask_inner u8, mask_sem u8 (= inner & label != -1);
 *   count i32[2] device out = {rows kept, rows inside the inner square};  ws i32[tl_tile_crop_ws_words(n)]. */
typedef struct tl_tile_box {
  float outer[4];
  double inner[4];
  double center[2];
  float half_inner;
} tl_tile_box;
int64_t tl_tile_crop_ws_words(int64_t n);
int tl_tile_crop(const float* xyz, const float* label, const float* feat, int64_t n, int F, const tl_tile_box* box,
                 float* coords, float* out_feat, int64_t* instance_labels, int64_t* semantic_labels,
                 uint8_t* mask_inner, uint8_t* mask_sem, int32_t* count, int32_t* ws, tl_stream_t stream);

/* ------------------------------------------------------------------ plot preparation
 * Global voxel down-sample and verticality feature, delegated by the reference to open3d 0.17.0
 * (VoxelDownSampleAndTrace; tree_learn/util/data_preparation.py:60-79) and jakteristics 0.5.1
 * (compute_features, data_preparation.py:82-88; called from generate_tiles, util/pipeline.py:40-65).
 *
 * tl_cell_keys: key[i] = pack(floor((p - min_bound) / cell) - base3), 21 bits per axis, z in the low bits;
 *   round_input != 0 rounds the coordinates to 2 decimals first (np.round(points, 2), data_preparation.py:62);
 *   base3 = HOST i64[3]; err i32[1] device out is set when a cell index leaves the 21-bit range.
 * tl_downsample_reduce: keys sorted ascending with the stable permutation `perm` (host code: torch.sort);
 *   per voxel, in input order and in double: mean of the 2-decimal-rounded points -> float32 -> rounded to 2
 *   decimals (util/pipeline.py:44-45); first_idx = smallest original index of the voxel (`idx_keep`);
 *   point2vox i64[n] = voxel row of every original point (the trace); n_voxels i64[1] device out.
 * tl_verticality: xyz_sorted f64[n,3] and keys sorted by a `radius`-sized cell key (tl_cell_keys, round_input = 0);
 *   extent2 = HOST i64[2] largest cell index in x and y; out f32[n] (sorted order) = 1 - |n_z| of the sample
 *   covariance of all points within `radius` (inclusive, the point itself included); NaN when fewer than 3. */
int tl_cell_keys(const double* xyz, int64_t n, double cell, double min_bound, const int64_t* base3, int round_input,
                 int64_t* keys, int32_t* err, tl_stream_t stream);
int64_t tl_downsample_ws_words(int64_t n);
int tl_downsample_reduce(const double* xyz, const int64_t* sorted_keys, const int64_t* perm, int64_t n, float* out_xyz,
                         int64_t* first_idx, int64_t* point2vox, int64_t* n_voxels, int32_t* ws, tl_stream_t stream);
/* Group means for `ensemble` (tree_learn/util/pipeline.py:113-141: pandas groupby(['x','y','z']).mean() over the rounded
 * coordinates): keys sorted ascending with the stable permutation `perm`; per group and column the double-precision sum of
 * the members in input order divided by their number.  src f32[n,C]; mean f64[n,C] capacity (first n_groups rows written);
 * first_idx i64 = smallest original index of each group; ws i32[tl_downsample_ws_words(n)].  Deterministic. */
int tl_group_mean(const float* src, int64_t n, int C, const int64_t* sorted_keys, const int64_t* perm, double* mean,
                  int64_t* first_idx, int64_t* n_groups, int32_t* ws, tl_stream_t stream);
int tl_verticality(const double* xyz_sorted, const int64_t* sorted_keys, int64_t n, double radius, const int64_t* extent2,
                   float* out, tl_stream_t stream);

/* ------------------------------------------------------------------ clustering
 * Replaces sklearn DBSCAN(eps, min_samples=2) in group_dbscan (tree_learn/util/pipeline.py:173-180):
 * connected components of the eps-graph on 2-D points; isolated points = -1; component labels
 * 0..K-1 ordered by the smallest point index of each component (sklearn's numbering).
 * xy f32[n,2]; labels i32[n] out; n_clusters i32[1] device out;
 * ws: tl_cluster_ws_bytes(n) bytes. */
int64_t tl_cluster_ws_bytes(int64_t n);
int tl_cluster_grid(const float* xy, int64_t n, double eps, int32_t* labels, int32_t* n_clusters,
                    void* ws, tl_stream_t stream);

/* HDBSCAN(min_cluster_size = m) of group_hdbscan (tree_learn/util/pipeline.py:184-191; sklearn: min_samples = m,
 * euclidean, Prim MST on mutual reachability, EOM).  Device stage: core distances (k-th neighbour incl. self) and
 * Prim's MST with sklearn's tie-breaking, fp64 on the fp32 points.  xy f32[n,2]; e_src/e_dst i32[n-1], e_w f64[n-1]
 * (edges in insertion order); core f64[n] or NULL; ws tl_hdbscan_ws_bytes(n). */
int64_t tl_hdbscan_ws_bytes(int64_t n);
int tl_hdbscan_mst(const float* xy, int64_t n, int min_samples, int32_t* e_src, int32_t* e_dst, double* e_w,
                   double* core, void* ws, tl_stream_t stream);
/* HOST stage (host pointers, no GPU work): edges -> single linkage -> condensed tree -> EOM -> labels i32[n]
 * (-1 noise, clusters numbered by ascending condensed-tree id like sklearn). */
int tl_hdbscan_labels_host(const int32_t* e_src, const int32_t* e_dst, const double* e_w, int64_t n,
                           int min_cluster_size, int32_t* labels);

/* The same device stage for large inputs (csrc/tl_hdbscan_grid.hip): core distances and the mutual-reachability MST through a
 * quadtree over a uniform cell grid -- grid k-NN for the core distances, Boruvka rounds for the tree -- instead of two O(n^2)
 * passes.  Same arithmetic (fp64 on the fp32 points), so core distances are bit-identical to tl_hdbscan_mst and the MST has the
 * same weight multiset; among equal-weight edges it takes the one that is smallest under (weight, smaller index, larger index)
 * instead of Prim's insertion order, and it returns the edges unordered (e_src < e_dst; pass them through
 * tl_hdbscan_prim_order_host before tl_hdbscan_labels_host).
 *   tl_hdbscan_grid_plan: picks the grid (bounding box; leaf level such that an occupied cell holds about eight points); ws of
 *     tl_hdbscan_grid_plan_ws_bytes(); SYNCHRONISES the stream (two small read-backs).
 *   tl_hdbscan_mst_grid: ws of tl_hdbscan_grid_ws_bytes(n, grid); SYNCHRONISES the stream once per Boruvka round (<= ~20).
 *     min_samples up to 4096 (reference util/pipeline.py:184-191 passes any tau_min): beyond 128 the k-best lists are heaps in the
 *     workspace -- size it with tl_hdbscan_grid_ws_bytes_k(n, grid, min_samples). */
typedef struct TlHdbGrid {
  double lo[2];     /* lower corner of the bounding box */
  double h;         /* leaf cell edge */
  int32_t levels;   /* leaf grid = 2^levels x 2^levels cells, 0..13 */
  int32_t reserved;
} TlHdbGrid;
int64_t tl_hdbscan_grid_plan_ws_bytes(void);
int tl_hdbscan_grid_plan(const float* xy, int64_t n, TlHdbGrid* grid, void* ws, tl_stream_t stream);
int64_t tl_hdbscan_grid_ws_bytes(int64_t n, const TlHdbGrid* grid);
int64_t tl_hdbscan_grid_ws_bytes_k(int64_t n, const TlHdbGrid* grid, int min_samples);
int tl_hdbscan_mst_grid(const float* xy, int64_t n, int min_samples, const TlHdbGrid* grid, int32_t* e_src, int32_t* e_dst,
                        double* e_w, double* core, void* ws, tl_stream_t stream);
/* HOST stage: spanning-tree edges in any order / orientation -> the order and orientation Prim's algorithm started at point 0
 * gives them (lightest frontier edge, smallest new index among equal weights, src = the end already in the tree), which is what
 * tl_hdbscan_labels_host's tie-breaking and cluster numbering key on.  All arrays i32/f64[n-1], host memory. */
int tl_hdbscan_prim_order_host(const int32_t* e_src, const int32_t* e_dst, const double* e_w, int64_t n, int32_t* o_src,
                               int32_t* o_dst, double* o_w);

/* ------------------------------------------------------------------ next-row helpers
 * k-NN majority vote: replaces KNeighborsClassifier(n_neighbors=k).fit(ref, labels).predict(query) in
 * assign_remaining_points_nearest_neighbor (tree_learn/util/pipeline.py:287-296).  ref_xyz f32[nr,3], ref_label i64[nr],
 * q_xyz f32[nq,3] -> out_label i64[nq]; k in {1,3,5}; ties -> smallest label; fp64 distances. */
int tl_knn_vote(const float* ref_xyz, const int64_t* ref_label, int64_t nr, const float* q_xyz, int64_t nq, int k,
                int64_t* out_label, tl_stream_t stream);
/* The same vote over a uniform cell grid (exact; bit-identical to tl_knn_vote): the caller bins the reference points
 * (cell = floor((p - lo) * (1.0f / h)) per axis in fp32, key = (cx * dims[1] + cy) * dims[2] + cz), sorts them by key and passes the
 * sorted coordinates / labels / ORIGINAL indices, the ncells distinct keys ascending and their row ranges cell_start[ncells + 1].
 * A query walks Chebyshev rings of cells until its k-th best distance beats anything an unvisited ring can hold; ties are broken
 * by (distance, original index) as in the brute-force form.  O(nq * points near the query) instead of O(nq * nr). */
int tl_knn_vote_grid(const float* ref_sorted_xyz, const int64_t* ref_sorted_label, const int64_t* ref_sorted_index, int64_t nr,
                     const int64_t* cell_keys, const int64_t* cell_start, int64_t ncells, const float lo[3], float h, const int32_t dims[3],
                     const float* q_xyz, int64_t nq, int k, int64_t* out_label, tl_stream_t stream);


/* ------------------------------------------------------------------ the whole eval-mode forward behind ONE call
 * Replaces, per batch of tiles, the body of `model(batch, return_loss=False)` of the reference's tile loop
 * (tree_learn/util/pipeline.py:86 -> tree_learn/model/tree_learn.py:75-103: voxelize :129-167, input conv :90, UBlock recursion
 * blocks.py:137-149, output_layer :93, forward_head :97-103): the yaml's configuration (use_feats = use_coords = False: all-ones
 * voxel features, configs/_modular/model.yaml:5-6) and the constructor's own defaults (use_feats = True, tree_learn.py:18: voxel-mean
 * features, tl_voxel_feats).  tl_forward enqueues everything the per-operator entry points
 * above would be called for -- tl_voxel_point_coords, tl_pyramid_build, tl_blk_build, tl_rulebooks_build, every tl_conv_fwd of the
 * U-Net in the pre-activated dataflow (BatchNorm + ReLU folded into producer epilogues / the staging prologue, residual adds and the
 * skip concat as views) and tl_head_mlp -- from C, so that the host cost of a forward no longer depends on the caller's interpreter.
 * It SYNCHRONISES `stream` twice (grid extent, level counts: two 16-B read-backs, as the per-operator path does).
 *
 * The network is described by plain structs of DEVICE pointers the caller keeps alive (weights in the tl_pack_weight layout, `frag` =
 * the tl_pack_weight_frag copy or NULL, eval-mode BatchNorms as scale / shift); nothing is copied.  All device memory of one forward
 * -- geometry and activations -- comes out of ONE caller-provided arena; if it is too small the call returns TL_ERR_ARENA with
 * args->needed_bytes set and nothing useful enqueued (call again with a larger arena).  Outputs are caller-provided and never alias
 * the arena, so the arena may serve the next forward on the same stream at once. */
#define TL_MAX_LEVELS 8
#define TL_ERR_ARENA (-4)        /* arena too small: args->needed_bytes */
#define TL_ERR_REACH_ZERO (-5)   /* a level's spatial shape or voxel set collapsed (spconv's "reach zero!!!", util/pipeline.py:91-97) */
#define TL_ERR_BLK (-7)          /* the block-local unit builder flagged skipped units (an internal assertion: cannot happen with the capacities tl_forward
                                    passes).  The flag comes home BEHIND the forward that raised it: tl_exec_check reports it for the forwards enqueued so
                                    far; a tl_forward that finds it set by its predecessor on the context returns it before enqueuing anything */
#define TL_ERR_EXTENT (-6)       /* the tile's voxel extent exceeds spatial_shape, a batch id is out of range, or a voxel coordinate leaves [0, 65536) */
typedef struct tl_affine { const float* scale; const float* shift; } tl_affine;
typedef struct tl_weight { const void* w; const void* frag; int32_t K, Cout, Cin, reserved; const void* x3; /* tl_pack_weight_x3 copy or NULL */ } tl_weight;
typedef struct tl_res_desc {      /* ResidualBlock, blocks.py:42-79 */
  tl_affine bn0; tl_weight w1; tl_affine bn3; tl_weight w2;
  tl_weight w1x1;                 /* i_branch: w == NULL = Identity (blocks.py:48-52) */
  tl_weight w1_half[2];           /* w1 split into its two input-channel halves (2C -> C decoder block on block-local rows), w == NULL = absent */
} tl_res_desc;
typedef struct tl_ublock_desc {   /* UBlock, blocks.py:81-149 (block_reps = 2, tree_learn.py:41) */
  int32_t C; int32_t deeper;
  tl_res_desc blocks[2];
  tl_affine bn_down; tl_weight wd;      /* BN, ReLU, SparseConv3d k2 s2 (blocks.py:102-110) */
  tl_affine bn_up; tl_weight wu;        /* BN, ReLU, SparseInverseConv3d k2 (blocks.py:116-123) */
  tl_res_desc tail[2];
  tl_affine bn_cat_l, bn_cat_r;         /* tail[0].bn0 split over the two halves of the skip concat (blocks.py:146) */
} tl_ublock_desc;
typedef struct tl_net_desc {
  int32_t dtype;                  /* TL_F32 | TL_BF16 | TL_F16 */
  int32_t num_levels;             /* 2 .. TL_MAX_LEVELS */
  float voxel_size;
  int32_t has_shape; int32_t spatial_shape[3];   /* tree_learn.py:86-87 override; has_shape = 0: the tile's own extent (:165) */
  int32_t blocked;                /* != 0: level 1 may run in the block-local order (16-bit, C = 32) */
  int32_t in_channels;            /* dim_coord + dim_feat of the input conv (tree_learn.py:38) */
  int32_t use_coords, use_feats;  /* tree_learn.py:152-155: both 0 = all-ones voxel features (no gather in the input conv); else args->point_feats is read */
  int32_t max_points_per_voxel;   /* tree_learn.py:22,141 (read when use_coords | use_feats) */
  int32_t reserved0;
  tl_weight w_in;                 /* input conv, tree_learn.py:37-39 */
  tl_ublock_desc u[TL_MAX_LEVELS];
  tl_affine out_bn;               /* output_layer, tree_learn.py:42 */
  const float* head_w1; const float* head_b1; const float* head_w2; const float* head_b2;   /* as tl_head_mlp takes them */
} tl_net_desc;
typedef struct tl_launch_rec {    /* one conv launch of a profiled forward */
  int32_t level, kind;            /* kind: 0 = SubM (27 taps), 1 = stride-2, 2 = inverse, 3 = 1x1, 4 = input conv */
  int32_t K, Cin, Cout, residual, esize, split_part, split_cin, in_prologue;
  int64_t n_out, n_in;
  float ms;
} tl_launch_rec;
typedef struct tl_forward_args {
  const float* xyz; const int64_t* batch_ids; int64_t N; int32_t B; int32_t reserved;
  const float* point_feats;       /* f32[N, in_channels - 3] (batch['input_feats']); may be NULL when the net uses neither coordinates nor features */
  void* arena; int64_t arena_bytes;
  float* backbone;                /* f32[N, C] or NULL */
  float* logits; float* offsets;  /* f32[N, 2], f32[N, 3] */
  tl_stream_t side_stream;        /* optional: the block-local unit builder runs there beside the other levels' rulebook kernels */
  int64_t needed_bytes;           /* out */
  int64_t level_n[TL_MAX_LEVELS]; /* out: active voxels per level */
  int32_t blocked_used;           /* out */
  int32_t launches;               /* out: tl_conv_fwd launches enqueued */
} tl_forward_args;
typedef struct tl_exec tl_exec;   /* host-side context of a caller thread / stream: pinned read-back words, two events, profile storage */
tl_exec* tl_exec_create(void);
void tl_exec_destroy(tl_exec* ex);
int tl_forward(tl_exec* ex, const tl_net_desc* net, tl_forward_args* args, tl_stream_t stream);
/* Waits for the unit-builder flag of the LAST tl_forward enqueued on this context (a 4-byte read-back behind its geometry kernels, not
 * behind its convs) and returns TL_ERR_BLK if that forward or an unreported earlier one raised it, else TL_OK.  Callers that need the
 * verdict for a particular tile call it before the next tl_forward on the context; a tile loop calls it once after its last tile. */
int tl_exec_check(tl_exec* ex);
/* Live per-launch timing of the NEXT tl_forward calls on this context (HIP events on the launch stream around every conv launch;
 * enable = 0 stops).  tl_exec_profile_read waits for the last profiled forward and returns its launches (<= cap; return value = their
 * number, negative = error). */
int tl_exec_profile(tl_exec* ex, int enable);
int tl_exec_profile_read(tl_exec* ex, tl_launch_rec* recs, int cap);


#ifdef __cplusplus
}
#endif
#endif /* TREELEARN_HIP_H */
