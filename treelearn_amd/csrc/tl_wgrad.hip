// Weight gradient of the sparse convolution (training step; reference: spconv's autograd behind SubMConv3d / SparseConv3d /
// SparseInverseConv3d.forward, reached from tools/training/train.py:40 `scaler.scale(loss).backward()`):
//     gW[k][co][ci] = sum over output rows o with table[k][o] >= 0 of  gout[o][co] * x[table[k][o]][ci]
// fp32 on the matrix cores with NO operand transposition: the 32x32x2 fp32 MFMA takes A[m][kk] and B[kk][n] as one float per
// lane with m = n = lane & 31 and kk = lane >> 5, so with kk = "which of two rows" both operands are plain 128-B row reads
// (A = 32 channels of gout[o], B = 32 channels of x[table[k][o]]): two (output row, input row) pairs per MFMA.
// Only PRESENT pairs are multiplied: a wave ballots the rulebook entries of 64 rows and walks the set bits two at a time
// (at level 1 only 5.5 of 27 taps are present per voxel, so this is ~5x less matrix work than the dense form).
// Work split: workgroup = (row chunk, tap, 64x64 block of [Cout x Cin]); each wave owns a quarter of the chunk and writes its
// partial [co][ci] tile to the workspace; tl_wgrad_reduce adds the partials in ascending chunk order -> deterministic.
// Replaces the round-1 "gather all [N, K, Cin] rows + one library GEMM" (6 GB of scratch per level-1 conv).
#include "tl_conv_internal.h"
#include "tl_f16_train.h"

namespace {

constexpr int kRowsPerWave = 4096;
constexpr int kWaves = 4;
constexpr int kSG = 256;           // bf16 kernel: rows whose present pairs are compacted into one list

// BF16: x and gout are bf16 (mixed-precision training); they are widened to fp32 in registers, the products and sums stay fp32
template <int NBO, int NBI, bool BF16>
__global__ void __launch_bounds__(kWaves * 64) k_wgrad(const void* __restrict__ x, int64_t x_ld, const void* __restrict__ g, int64_t g_ld,
                                                       const int32_t* __restrict__ table, int64_t n_out, int64_t n_in, int K, int Cin, int Cout,
                                                       int nbi_blocks, float* __restrict__ ws) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  const int k = blockIdx.y;
  const int bo = blockIdx.z / nbi_blocks, bi = blockIdx.z % nbi_blocks;
  const int co0 = bo * (NBO * 32), ci0 = bi * (NBI * 32);
  const int64_t part = (int64_t)blockIdx.x * kWaves + wv;
  const int64_t r_begin = part * kRowsPerWave, r_end = min(n_out, r_begin + kRowsPerWave);

  constexpr int EB = BF16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, (int)min((int64_t)0x7FFFFFFF, n_in * x_ld * EB), 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g), 0, (int)min((int64_t)0x7FFFFFFF, n_out * g_ld * EB), 0x00020000);
  auto ld = [&](const __amdgpu_buffer_rsrc_t& r, unsigned off) __attribute__((always_inline)) -> float {
    if constexpr (BF16) return __uint_as_float((uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, (int)off, 0, 0) << 16);
    else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
  };
  unsigned offx[NBI], offg[NBO];                 // per-lane channel offsets (bytes), out of range -> 0xFFFFFFFF (reads as 0)
#pragma unroll
  for (int b = 0; b < NBI; ++b) offx[b] = (ci0 + b * 32 + fi < Cin) ? (unsigned)((ci0 + b * 32 + fi) * EB) : 0xFFFFFFFFu;
#pragma unroll
  for (int b = 0; b < NBO; ++b) offg[b] = (co0 + b * 32 + fi < Cout) ? (unsigned)((co0 + b * 32 + fi) * EB) : 0xFFFFFFFFu;

  f32x16 acc[NBO][NBI];
#pragma unroll
  for (int a = 0; a < NBO; ++a)
#pragma unroll
    for (int b = 0; b < NBI; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  constexpr int PF = 8;                          // (row, row) pairs whose loads are in flight together
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 64) {
    const int64_t row = r0 + lane;
    int idx = -1;
    if (row < r_end) idx = table ? table[(int64_t)k * n_out + row] : (int)row;
    unsigned long long m = __ballot(idx >= 0);
    while (m) {
      float gv[PF][NBO], xv[PF][NBI];
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        // two present rows of this 64-row group (or one and nothing): lane half fh takes the fh-th of them
        int b1 = -1, b2 = -1;
        if (m) { b1 = __builtin_ctzll(m); m &= m - 1; }
        if (m) { b2 = __builtin_ctzll(m); m &= m - 1; }
        const int i1 = b1 >= 0 ? __builtin_amdgcn_readlane(idx, b1) : -1;
        const int i2 = b2 >= 0 ? __builtin_amdgcn_readlane(idx, b2) : -1;
        const int bsel = fh ? b2 : b1, isel = fh ? i2 : i1;
        const unsigned gbase = bsel >= 0 ? (unsigned)((r0 + bsel) * g_ld * EB) : 0xFFFFFFFFu;
        const unsigned xbase = isel >= 0 ? (unsigned)((int64_t)isel * x_ld * EB) : 0xFFFFFFFFu;
#pragma unroll
        for (int b = 0; b < NBO; ++b)
          gv[u][b] = ld(rg, (gbase | offg[b]) == 0xFFFFFFFFu ? 0xFFFFFFFFu : gbase + offg[b]);
#pragma unroll
        for (int b = 0; b < NBI; ++b)
          xv[u][b] = ld(rx, (xbase | offx[b]) == 0xFFFFFFFFu ? 0xFFFFFFFFu : xbase + offx[b]);
      }
#pragma unroll
      for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int a = 0; a < NBO; ++a)
#pragma unroll
          for (int b = 0; b < NBI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[u][a], xv[u][b], acc[a][b], 0, 0, 0);
    }
  }

  // partial tile -> workspace [part][K][Cout][Cin]
  float* wp = ws + ((part * K + k) * (int64_t)Cout) * Cin;
#pragma unroll
  for (int a = 0; a < NBO; ++a)
#pragma unroll
    for (int b = 0; b < NBI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh, ci = ci0 + b * 32 + fi;
        if (co < Cout && ci < Cin) wp[(int64_t)co * Cin + ci] = acc[a][b][r];
      }
}

// bf16 inputs on the bf16 matrix cores (mixed-precision training): 16 (output row, input row) pairs per 32x32x16 MFMA instead of
// two per fp32 MFMA.  The contraction index is the PAIR, so lane (m, h) needs 8 different rows at one column -- not something a
// row-major gather delivers.  Hence: the wave compacts the present pairs of a 64-row group into a wave-private LDS list
// (rank = popcount of the lower lanes' ballot bits), fetches 16 pair rows of gout and of x at a time with full-row 16-B loads
// (NB*4 lanes per row), parks them row-major in LDS and reads the operand columns back as 2-byte elements (32 consecutive
// channels of one row per half-wave: conflict-free).  Accumulation stays fp32; partial tiles and the ordered reduction are those
// of the fp32 kernel.  NBO, NBI in {1, 2, 3}; channel counts must be multiples of 8 (16-B pieces).
// TR: the operand columns come back from LDS through ds_read_b64_tr_b16 (gfx950's transposing read: a 16-lane group reads a
// [4 pairs][16 channels] block, lane i supplying the address of 8-byte chunk i = (pair i >> 2, channels 4 (i & 3) ..) and
// receiving channel i of the four pairs): 2 reads per 32-channel operand block instead of 8 two-byte reads + 4 packs.
template <int NBO, int NBI, bool TR = true>
__global__ void __launch_bounds__(kWaves * 64) k_wgrad_bf16(const uint16_t* __restrict__ x, int64_t x_ld, const uint16_t* __restrict__ g, int64_t g_ld,
                                                            const int32_t* __restrict__ table, int64_t n_out, int64_t n_in, int K, int Cin, int Cout,
                                                            int nbi_blocks, float* __restrict__ ws) {
  // staging pattern: LG (LX) lanes cover one row of the gout (x) block with 16-B pieces; a 96-channel block (NB = 3) takes 16 lanes per
  // row like a 128-channel one, its last four pieces masked (they read as zeros and land in the row's padding)
  constexpr int LG = NBO == 3 ? 16 : NBO * 4, LX = NBI == 3 ? 16 : NBI * 4, RG = 64 / LG, RX = 64 / LX;       // lanes per row, rows per load instruction
  constexpr int GP = LG * 8 + 8, XP = LX * 8 + 8;            // LDS row pitch in elements (+16 B: the 16-B row writes of 8 rows per instruction spread over the banks)
  constexpr int kStage = kWaves * 16 * (GP + XP) * 2 + kWaves * kSG * 8, kRed = NBO * 32 * (NBI * 32 + 1) * 4;
  __shared__ __attribute__((aligned(16))) char smem_w[kStage > kRed ? kStage : kRed];
  uint16_t (*Gs)[16][GP] = reinterpret_cast<uint16_t (*)[16][GP]>(smem_w);
  uint16_t (*Xs)[16][XP] = reinterpret_cast<uint16_t (*)[16][XP]>(smem_w + kWaves * 16 * GP * 2);
  int2 (*Ls)[kSG] = reinterpret_cast<int2 (*)[kSG]>(smem_w + kWaves * 16 * (GP + XP) * 2);       // compacted (row, input row) pairs of the current super-group of rows
  float (*Rs)[NBI * 32 + 1] = reinterpret_cast<float (*)[NBI * 32 + 1]>(smem_w);                  // after the main loop: the workgroup's tile sum
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  const int k = blockIdx.y;
  const int bo = blockIdx.z / nbi_blocks, bi = blockIdx.z % nbi_blocks;
  const int co0 = bo * (NBO * 32), ci0 = bi * (NBI * 32);
  const int64_t part = (int64_t)blockIdx.x * kWaves + wv;
  const int64_t r_begin = part * kRowsPerWave, r_end = min(n_out, r_begin + kRowsPerWave);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, (int)min((int64_t)0x7FFFFFFF, n_in * x_ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g), 0, (int)min((int64_t)0x7FFFFFFF, n_out * g_ld * 2), 0x00020000);
  const int g_piece = lane % LG, g_row = lane / LG, x_piece = lane % LX, x_row = lane / LX;
  const unsigned g_coff = (g_piece < NBO * 4 && co0 + g_piece * 8 < Cout) ? (unsigned)((co0 + g_piece * 8) * 2) : 0xFFFFFFFFu;
  const unsigned x_coff = (x_piece < NBI * 4 && ci0 + x_piece * 8 < Cin) ? (unsigned)((ci0 + x_piece * 8) * 2) : 0xFFFFFFFFu;

  f32x16 acc[NBO][NBI];
#pragma unroll
  for (int a = 0; a < NBO; ++a)
#pragma unroll
    for (int b = 0; b < NBI; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // 16 pair rows of gout and of x, batch starting at list position p0 -> registers (absent pairs / channels past the end read as zeros)
  constexpr int NG = 16 / RG, NX = 16 / RX;
  auto fetch = [&](int p0, int cnt, u32x4 (&gq)[NG], u32x4 (&xq)[NX]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int p = p0 + g_row + i * RG;
      const int2 e = Ls[wv][p < cnt ? p : 0];
      const unsigned base = (p < cnt && g_coff != 0xFFFFFFFFu) ? (unsigned)((r_begin + e.x) * g_ld * 2) + g_coff : 0xFFFFFFFFu;
      gq[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)base, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int p = p0 + x_row + i * RX;
      const int2 e = Ls[wv][p < cnt ? p : 0];
      const unsigned base = (p < cnt && x_coff != 0xFFFFFFFFu) ? (unsigned)((int64_t)e.y * x_ld * 2) + x_coff : 0xFFFFFFFFu;
      xq[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)base, 0, 0));
    }
  };
  // one batch: registers -> LDS (row-major), operand columns back (transposed), NBO x NBI MFMAs
  auto contract = [&](const u32x4 (&gq)[NG], const u32x4 (&xq)[NX]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NG; ++i) *reinterpret_cast<u32x4*>(&Gs[wv][g_row + i * RG][g_piece * 8]) = gq[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) *reinterpret_cast<u32x4*>(&Xs[wv][x_row + i * RX][x_piece * 8]) = xq[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // operand columns: lane (m = fi, h = fh) takes pairs 8h .. 8h+7 of channel m of its block
    u32x4 A[NBO], B[NBI];
    if constexpr (TR) {
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      typedef __attribute__((address_space(3))) s16x4* lds4;
      const int ti = lane & 15, tg = (lane >> 4) & 1;                       // lane in its 16-lane group; which 16-channel half of the block
      const int prow = 8 * fh + (ti >> 2), pcol = 16 * tg + 4 * (ti & 3);
#pragma unroll
      for (int a = 0; a < NBO; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(&Gs[wv][prow + 4 * q][a * 32 + pcol])));
          A[a][2 * q] = v[0]; A[a][2 * q + 1] = v[1];
        }
#pragma unroll
      for (int b = 0; b < NBI; ++b)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(&Xs[wv][prow + 4 * q][b * 32 + pcol])));
          B[b][2 * q] = v[0]; B[b][2 * q + 1] = v[1];
        }
    } else {
#pragma unroll
      for (int a = 0; a < NBO; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          A[a][t] = (uint32_t)Gs[wv][8 * fh + 2 * t][a * 32 + fi] | ((uint32_t)Gs[wv][8 * fh + 2 * t + 1][a * 32 + fi] << 16);
#pragma unroll
      for (int b = 0; b < NBI; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          B[b][t] = (uint32_t)Xs[wv][8 * fh + 2 * t][b * 32 + fi] | ((uint32_t)Xs[wv][8 * fh + 2 * t + 1][b * 32 + fi] << 16);
    }
#pragma unroll
    for (int a = 0; a < NBO; ++a)
#pragma unroll
      for (int b = 0; b < NBI; ++b)
        acc[a][b] = h16_mfma(A[a], B[b], acc[a][b]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  // Rows are walked in super-groups of kSG: their present pairs are compacted into ONE wave-private list (fewer partly filled
  // 16-pair batches than per 64-row group), and the batches run through a two-deep register pipeline: while batch b is staged
  // through LDS and contracted, the rows of batches b+1 and b+2 are already in flight (the chain global load -> LDS -> MFMA of one
  // batch is otherwise a full memory latency per 16 pairs).
  for (int64_t r0 = r_begin; r0 < r_end; r0 += kSG) {
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < kSG / 64; ++q) {
      const int64_t row = r0 + q * 64 + lane;
      int idx = -1;
      if (row < r_end) idx = table ? table[(int64_t)k * n_out + row] : (int)row;
      const unsigned long long m = __ballot(idx >= 0);
      if (idx >= 0) Ls[wv][cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = make_int2((int)(row - r_begin), idx);
      cnt += __builtin_popcountll(m);
    }
    if (cnt == 0) continue;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // branch-free: every step issues its prefetch and its contraction (past the end of the list all offsets are out of range: no
    // memory access, zeros back, a zero contribution), so the loop body is straight-line code and the waits for a batch stay
    // counted (vmcnt(N)) -- with `if (more) fetch(...)` the control-flow join drained the younger prefetches on every batch
    u32x4 g0[NG], x0[NX], g1[NG], x1[NX];
    fetch(0, cnt, g0, x0);
    fetch(16, cnt, g1, x1);
    for (int p0 = 0; p0 < cnt; p0 += 32) {
      {
        u32x4 gc[NG], xc[NX];
#pragma unroll
        for (int i = 0; i < NG; ++i) gc[i] = g0[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = x0[i];
        fetch(p0 + 32, cnt, g0, x0);
        contract(gc, xc);
      }
      {
        u32x4 gc[NG], xc[NX];
#pragma unroll
        for (int i = 0; i < NG; ++i) gc[i] = g1[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = x1[i];
        fetch(p0 + 48, cnt, g1, x1);
        contract(gc, xc);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                    // the list is rebuilt by the next super-group
  }

  // the four waves' tiles are added in wave order through LDS (deterministic) -> ONE partial tile per workgroup:
  // workspace [row chunk][K][Cout][Cin], a quarter of the traffic and of the reduction kernel's work
  __syncthreads();
  float* wp = ws + (((int64_t)blockIdx.x * K + k) * (int64_t)Cout) * Cin;
  for (int w = 0; w < kWaves; ++w) {
    if (wv == w) {
#pragma unroll
      for (int a = 0; a < NBO; ++a)
#pragma unroll
        for (int b = 0; b < NBI; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int col = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh, cil = b * 32 + fi;
            float v = acc[a][b][r];
            if (w > 0) v += Rs[col][cil];
            if (w < kWaves - 1) Rs[col][cil] = v;
            else if (co0 + col < Cout && ci0 + cil < Cin) wp[(int64_t)(co0 + col) * Cin + ci0 + cil] = v;
          }
    }
    __syncthreads();
  }
}

// The same contraction with the gout rows SHARED: both operands of the kernel above are gathered per pair (4 KB of rows per four
// MFMAs at 64 x 64 -- twice the forward conv's ratio, and the L1 gather rate is what bounds it), yet the gout rows of a row range
// are the same for all 27 taps.  Here a workgroup = (row chunk, group of 4 T taps, channel block): the four waves walk the SAME rows
// in super-groups of R, the gout tile of a super-group is staged in LDS once (coalesced, prefetched into registers one super-group
// ahead), wave w contracts taps w, w + 4, ... against it -- the A fragments come from the shared tile through ds_read_b64_tr_b16 at
// the pairs' own rows (every lane supplies its row address, so the compacted pair list needs no copy of gout), only the x rows are
// gathered per pair.  gout traffic drops from one row per pair to 1 / (taps per workgroup x tap density) rows per pair; every wave
// owns whole taps, so its tiles go to the workspace without the cross-wave reduction.
// R = rows per super-group; DENSE: no pair compaction (absent pairs multiply zeros); MODE (developer A/B, 1 ships): 0 = prefetches under
// `if (more)`, 1 = branch-free batch pipeline, 2 = 1 + the next super-group's tile / rulebook entries requested behind the first x rows
template <int NBO, int NBI, int T, int R, bool DENSE = false, int MODE = 1>
__global__ void __launch_bounds__(kWaves * 64) k_wgrad_bf16s(const uint16_t* __restrict__ x, int64_t x_ld, const uint16_t* __restrict__ g, int64_t g_ld,
                                                             const int32_t* __restrict__ table, int64_t n_out, int64_t n_in, int K, int Cin, int Cout,
                                                             int nbi_blocks, int chunk_rows, float* __restrict__ ws) {
  constexpr int LG = NBO == 3 ? 16 : NBO * 4, LX = NBI == 3 ? 16 : NBI * 4, RX = 64 / LX;
  constexpr int GP = LG * 8 + 8, XP = LX * 8 + 8;
  constexpr int NPT = R * LG / (kWaves * 64);                                  // 16-B pieces of the gout tile per thread
  constexpr int NX = 16 / RX;
  __shared__ __attribute__((aligned(16))) uint16_t Gt[R + 1][GP];              // row R stays zero: the row of absent pairs
  __shared__ __attribute__((aligned(16))) uint16_t Xs[kWaves][16][XP];
  __shared__ int2 Ls[kWaves][R];                                               // (row in the super-group, input row) of the present pairs of the wave's current tap
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, tid = threadIdx.x;
  const int fi = lane & 31, fh = lane >> 5;
  const int bo = blockIdx.z / nbi_blocks, bi = blockIdx.z % nbi_blocks;
  const int co0 = bo * (NBO * 32), ci0 = bi * (NBI * 32);
  const int64_t c_begin = (int64_t)blockIdx.x * chunk_rows, c_end = min(n_out, c_begin + (int64_t)chunk_rows);
  const int k_first = blockIdx.y * (kWaves * T) + wv;                          // this wave's taps: k_first + kWaves * t
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, (int)min((int64_t)0x7FFFFFFF, n_in * x_ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g), 0, (int)min((int64_t)0x7FFFFFFF, n_out * g_ld * 2), 0x00020000);
  const int x_piece = lane % LX, x_row = lane / LX;
  const unsigned x_coff = (x_piece < NBI * 4 && ci0 + x_piece * 8 < Cin) ? (unsigned)((ci0 + x_piece * 8) * 2) : 0xFFFFFFFFu;

  f32x16 acc[T][NBO][NBI];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int a = 0; a < NBO; ++a)
#pragma unroll
      for (int b = 0; b < NBI; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][a][b][i] = 0.f;

  // the gout tile of the super-group starting at row r0 -> registers (rows past the chunk / channels past the block read as zeros)
  auto tile_load = [&](int64_t r0, u32x4 (&gq)[NPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int q = tid + i * (kWaves * 64), row = q / LG, piece = q % LG;
      const bool ok = r0 + row < c_end && piece < NBO * 4 && co0 + piece * 8 < Cout;
      const unsigned base = ok ? (unsigned)((r0 + row) * g_ld * 2) + (unsigned)((co0 + piece * 8) * 2) : 0xFFFFFFFFu;
      gq[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)base, 0, 0));
    }
  };
  auto tile_store = [&](const u32x4 (&gq)[NPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int q = tid + i * (kWaves * 64);
      *reinterpret_cast<u32x4*>(&Gt[q / LG][(q % LG) * 8]) = gq[i];
    }
  };
  auto fetch = [&](int p0, int cnt, u32x4 (&xq)[NX]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int p = p0 + x_row + i * RX;
      const int2 e = Ls[wv][p < cnt ? p : 0];
      const unsigned base = (p < cnt && e.y >= 0 && x_coff != 0xFFFFFFFFu) ? (unsigned)((int64_t)e.y * x_ld * 2) + x_coff : 0xFFFFFFFFu;
      xq[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)base, 0, 0));
    }
  };
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) s16x4* lds4;
  const int ti = lane & 15, tg = (lane >> 4) & 1;
  const int prow = 8 * fh + (ti >> 2), pcol = 16 * tg + 4 * (ti & 3);
  auto contract = [&](int p0, int cnt, const u32x4 (&xq)[NX], f32x16 (&ac)[NBO][NBI]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NX; ++i) *reinterpret_cast<u32x4*>(&Xs[wv][x_row + i * RX][x_piece * 8]) = xq[i];
    int grow[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { const int pp = p0 + prow + 4 * q; grow[q] = DENSE ? pp : (pp < cnt ? Ls[wv][pp].x : R); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32x4 A[NBO], B[NBI];
#pragma unroll
    for (int a = 0; a < NBO; ++a)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(&Gt[grow[q]][a * 32 + pcol])));
        A[a][2 * q] = v[0]; A[a][2 * q + 1] = v[1];
      }
#pragma unroll
    for (int b = 0; b < NBI; ++b)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(&Xs[wv][prow + 4 * q][b * 32 + pcol])));
        B[b][2 * q] = v[0]; B[b][2 * q + 1] = v[1];
      }
#pragma unroll
    for (int a = 0; a < NBO; ++a)
#pragma unroll
      for (int b = 0; b < NBI; ++b)
        ac[a][b] = h16_mfma(A[a], B[b], ac[a][b]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  for (int c = tid; c < GP; c += kWaves * 64) Gt[R][c] = 0;
  {
    u32x4 g0[NPT];
    tile_load(c_begin, g0);
    tile_store(g0);
  }
  __syncthreads();
  constexpr int D = 2;                                     // batches of x rows in flight per wave
  // rulebook entries of the first super-group, for every tap of the wave
  int idx_n[T][R / 64];
  auto table_load = [&](int64_t r0) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int k = k_first + kWaves * t;
#pragma unroll
      for (int q = 0; q < R / 64; ++q) {
        const int64_t row = r0 + q * 64 + lane;
        idx_n[t][q] = (k < K && row < c_end) ? (table ? table[(int64_t)k * n_out + row] : (int)row) : -1;
      }
    }
  };
  if (MODE == 2) table_load(c_begin);
  for (int64_t r0 = c_begin; r0 < c_end; r0 += R) {
    u32x4 gn[NPT];
    const bool more = r0 + R < c_end;
    if (MODE != 2) table_load(r0);
    int idx_c[T][R / 64];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int q = 0; q < R / 64; ++q) idx_c[t][q] = idx_n[t][q];
    bool ahead = MODE != 2;                                // next super-group's gout tile and rulebook entries requested?
    if (MODE != 2 && more) tile_load(r0 + R, gn);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int k = k_first + kWaves * t;
      int cnt = 0;
      if (k < K) {
#pragma unroll
        for (int q = 0; q < R / 64; ++q) {
          const int idx = idx_c[t][q];
          if constexpr (DENSE) {
            Ls[wv][q * 64 + lane] = make_int2(q * 64 + lane, idx);
            cnt += 64;
          } else {
            const unsigned long long m = __ballot(idx >= 0);
            if (idx >= 0) Ls[wv][cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = make_int2(q * 64 + lane, idx);
            cnt += __builtin_popcountll(m);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // branch-free pipeline: every step issues its prefetch (past the end of the list every lane's offset is out of range: no memory
      // access, zeros back), so the loop body is straight-line code and the waits for a batch stay counted (vmcnt(N)) instead of
      // draining the younger prefetches at a control-flow join
      u32x4 xb[D][NX];
#pragma unroll
      for (int d = 0; d < D; ++d) if (MODE != 0 || 16 * d < cnt) fetch(16 * d, cnt, xb[d]);
      if (!ahead) {                                        // behind this tap's first x rows in the (in-order) load queue
        ahead = true;
        if (more) { tile_load(r0 + R, gn); table_load(r0 + R); }
      }
      for (int p0 = 0; p0 < cnt; p0 += 16 * D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          u32x4 xc[NX];
#pragma unroll
          for (int i = 0; i < NX; ++i) xc[i] = xb[d][i];
          if (MODE != 0 || p0 + 16 * (d + D) < cnt) fetch(p0 + 16 * (d + D), cnt, xb[d]);
          if (MODE != 0 || p0 + 16 * d < cnt) contract(p0 + 16 * d, cnt, xc, acc[t]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                    // the list is rebuilt by the next tap
    }
    __syncthreads();                                       // every wave is done with this tile
    if (more) tile_store(gn);
    __syncthreads();
  }
  // every wave owns whole taps: its tiles are the workgroup's partials -- workspace [row chunk][K][Cout][Cin]
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int k = k_first + kWaves * t;
    if (k >= K) continue;
    float* wp = ws + (((int64_t)blockIdx.x * K + k) * (int64_t)Cout) * Cin;
#pragma unroll
    for (int a = 0; a < NBO; ++a)
#pragma unroll
      for (int b = 0; b < NBI; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh, ci = ci0 + b * 32 + fi;
          if (co < Cout && ci < Cin) wp[(int64_t)co * Cin + ci] = acc[t][a][b][r];
        }
  }
}

__global__ void k_wgrad_reduce(const float* __restrict__ ws, int64_t nparts, int64_t per, float* __restrict__ gw) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int64_t p = 0; p < nparts; ++p) s += ws[p * per + e];          // ascending chunk order: deterministic
    gw[e] = s;
  }
}

}  // namespace

static int g_wgrad_var = 0, g_wgrad_chunk = 0;   // developer A/B: kernel variant / fixed chunk rows
static int g_wgrad_shared = 1;             // developer A/B (tl_dev_wgrad_mode bit 1): 0 = per-tap workgroups that gather gout per pair
static int g_wgrad_bf16_mfma = 1;          // developer A/B (tl_dev_wgrad_mode): 0 = bf16 inputs through the fp32-MFMA kernel

// tl_wgrad_dense.hip: the dense-over-taps form of the 27-tap convs of the big levels
int tl_wgrad_dense_slots(int64_t n_out, int K, int Cin, int Cout);
int tl_launch_wgrad_dense(const uint16_t* x, int64_t x_ld, const uint16_t* g, int64_t g_ld, const int32_t* table, int64_t n_out, int64_t n_in, int K, int Cin,
                          int Cout, float* gw, float* ws, hipStream_t s, int ref_layout);
int tl_launch_wgrad_reduce(const float* ws, int64_t nparts, int64_t per, float* gw, hipStream_t s, int K = 1, int Cout = 0, int Cin = 0, int ref_layout = 0);
// tl_linear_small.hip: K = 1 with <= 4 output channels (the heads' output Linears)
// tl_wgrad_rows.hip: K = 1 (1x1 convs, the heads' hidden Linears) as a row-streaming GEMM
int tl_wgrad_rows_parts(int64_t n, int Cin, int Cout);
int tl_launch_wgrad_rows(const uint16_t* x, int64_t x_ld, const uint16_t* g, int64_t g_ld, int64_t n, int Cin, int Cout, float* gw, float* ws, hipStream_t s);
int tl_wgrad_in4_parts(int64_t n, int K, int Cin, int Cout);
int tl_launch_wgrad_in4(const uint16_t* x, int64_t x_ld, const uint16_t* g, int64_t g_ld, const int32_t* table, int64_t n, int64_t n_in, float* gw, float* ws,
                        hipStream_t s, int ref_layout);
int64_t tl_wgrad_tinycout_parts(int64_t n);
int tl_launch_wgrad_tinycout(const void* x, int64_t x_ld, const void* g, int64_t g_ld, int dtype, int64_t n, int Cin, int Cout, float* gw, float* ws, hipStream_t s);

extern "C" {

#ifdef TL_DEV
int tl_dev_wgrad_mode(int mode) {
  g_wgrad_bf16_mfma = mode & 1; g_wgrad_shared = !(mode & 2); g_wgrad_var = (mode >> 8) & 15;
  g_wgrad_chunk = ((mode >> 12) & 3) ? 2048 << ((mode >> 12) & 3) : 0;
  return TL_OK;
}
#endif

int64_t tl_conv_wgrad_ws_floats(int64_t n_out, int K, int Cin, int Cout) {
  int64_t nparts = tl_cdiv(n_out, (int64_t)kRowsPerWave * kWaves) * kWaves;
  const int64_t dense = tl_wgrad_dense_slots(n_out, K, Cin, Cout);
  if (dense > nparts) nparts = dense;
  if (K == 1 && Cout <= 4 && tl_wgrad_tinycout_parts(n_out) > nparts) nparts = tl_wgrad_tinycout_parts(n_out);
  if (K == 1 && tl_wgrad_rows_parts(n_out, Cin, Cout) > nparts) nparts = tl_wgrad_rows_parts(n_out, Cin, Cout);
  if (tl_wgrad_in4_parts(n_out, K, Cin, Cout) > nparts) nparts = tl_wgrad_in4_parts(n_out, K, Cin, Cout);
  return nparts * K * Cout * Cin;
}

static int wgrad_impl(const void* x, int64_t x_ld, const void* gout, int64_t g_ld, int dtype, const int32_t* table, int64_t n_out, int64_t n_in, int K,
                      int Cin, int Cout, float* gw, float* ws, tl_stream_t stream, int ref_layout) {
  if (!x || !gout || !gw || !ws || n_out <= 0 || n_in <= 0 || K <= 0 || Cin <= 0 || Cout <= 0 || x_ld < Cin || g_ld < Cout) return TL_ERR_ARG;
  if (!table && K != 1) return TL_ERR_ARG;
  if (dtype != TL_F32 && dtype != TL_BF16) return TL_ERR_ARG;
  const int eb = dtype == TL_BF16 ? 2 : 4;
  if (n_in * x_ld * eb > 0x7FFFFFFFll || n_out * g_ld * eb > 0x7FFFFFFFll) return TL_ERR_UNSUPPORTED;     // 32-bit buffer offsets
  hipStream_t s = tl_s(stream);
  const int64_t nchunks = tl_cdiv(n_out, (int64_t)kRowsPerWave * kWaves);
  int64_t nchunks_used = nchunks;
  const bool bf16_mfma = dtype == TL_BF16 && Cout % 8 == 0 && Cin % 8 == 0 && x_ld % 8 == 0 && g_ld % 8 == 0 && ((uintptr_t)x) % 16 == 0 &&
                         ((uintptr_t)gout) % 16 == 0 && g_wgrad_bf16_mfma;
  if (K == 1 && !table && Cout <= 4 && n_in == n_out) {
    const int rc = tl_launch_wgrad_tinycout(x, x_ld, gout, g_ld, dtype, n_out, Cin, Cout, gw, ws, s);
    if (rc != TL_ERR_UNSUPPORTED) return rc;
  }
  if (dtype == TL_BF16 && K == 27 && Cin == 4 && Cout == 32 && table) {
    const int rc = tl_launch_wgrad_in4((const uint16_t*)x, x_ld, (const uint16_t*)gout, g_ld, table, n_out, n_in, gw, ws, s, ref_layout);
    if (rc != TL_ERR_UNSUPPORTED) return rc;
  }
  if (bf16_mfma && K == 1 && !table && n_in == n_out) {
    const int rc = tl_launch_wgrad_rows((const uint16_t*)x, x_ld, (const uint16_t*)gout, g_ld, n_out, Cin, Cout, gw, ws, s);
    if (rc != TL_ERR_UNSUPPORTED) return rc;
  }
  if (bf16_mfma && table) {
    const int rc = tl_launch_wgrad_dense((const uint16_t*)x, x_ld, (const uint16_t*)gout, g_ld, table, n_out, n_in, K, Cin, Cout, gw, ws, s, ref_layout);
    if (rc != TL_ERR_UNSUPPORTED) return rc;
  }
  if (bf16_mfma) {
    // bf16 matrix cores: blocks of at most 64 x 64 channels (partially filled blocks are masked); on big levels whose widths are
    // multiples of 96 (level 3: 96 -> 96, 192 -> 96) 96 x 96 blocks -- ONE 3 x 3 block per 96 x 96 (9 MFMAs per 16 pairs, rows
    // gathered once) instead of four 2 x 2 blocks (16 MFMAs, rows gathered four times): 2.0 -> 1.08 ms.  Mixed 2/3 shapes and small
    // levels lose (one wave per SIMD, fewer workgroups) and keep the 64-wide blocks.
    const bool wide = Cout % 96 == 0 && Cin % 96 == 0 && n_out >= 100000;
    const int to = wide ? 3 : (Cout > 32 ? 2 : 1), ti = wide ? 3 : (Cin > 32 ? 2 : 1);
    const int nbo = (int)tl_cdiv(Cout, 32 * to), nbi = (int)tl_cdiv(Cin, 32 * ti);
    const uint16_t* xb = (const uint16_t*)x; const uint16_t* gb = (const uint16_t*)gout;
    if (K == 27 && to >= 2 && ti >= 2 && g_wgrad_shared) {
      // shared-gout form (27-tap convs with both sides >= 64 channels; 32-channel sides and the 8-tap down / up convs measured
      // slower in it): workgroup = (row chunk, 4 taps, block); chunks of 4 096 .. 16 384 rows (the workspace holds one partial per
      // 4 096 rows), the smallest that still gives every CU a few workgroups
      int64_t chunk = 16384;
      if (g_wgrad_chunk) chunk = g_wgrad_chunk;
      else while (chunk > 4096 && tl_cdiv(n_out, chunk) * tl_cdiv(K, kWaves) * nbo * nbi < 2048) chunk >>= 1;
      nchunks_used = tl_cdiv(n_out, chunk);
#define TL_WS(O_, I_, T_, R_, ...)                                                                                          \
  k_wgrad_bf16s<O_, I_, T_, R_, ##__VA_ARGS__><<<dim3((unsigned)nchunks_used, (unsigned)tl_cdiv(K, kWaves * T_), (unsigned)(nbo * nbi)), kWaves * 64, 0, s>>>( \
      xb, x_ld, gb, g_ld, table, n_out, n_in, K, Cin, Cout, nbi, (int)chunk, ws)
#ifdef TL_DEV
      const int v = g_wgrad_var;                               // tools/dev_wgrad_var.py
      if (v && to == 2) {
        if (v == 1) TL_WS(2, 2, 1, 128, false, 0); else if (v == 2) TL_WS(2, 2, 1, 128, false, 2); else if (v == 3) TL_WS(2, 2, 1, 128, true, 0);
        else if (v == 4) TL_WS(2, 2, 1, 128, true, 1); else if (v == 5) TL_WS(2, 2, 1, 128, true, 2); else if (v == 6) TL_WS(2, 2, 2, 128, true, 1);
        else TL_WS(2, 2, 1, 256, true, 1);
      } else if (v && to == 3) {
        if (v == 1) TL_WS(3, 3, 1, 128, false, 1); else if (v == 3) TL_WS(3, 3, 1, 128, true, 0); else TL_WS(3, 3, 1, 128, false, 0);
      } else
#endif
      if (to == 3) TL_WS(3, 3, 1, 128, true, 1);              // level 3 (24 of 27 taps present): no pair compaction
      else TL_WS(2, 2, 1, 128, false, 1);
#undef TL_WS
    } else {
    const dim3 grid((unsigned)nchunks, (unsigned)K, (unsigned)(nbo * nbi));
    if (to == 3 && ti == 3) k_wgrad_bf16<3, 3><<<grid, kWaves * 64, 0, s>>>(xb, x_ld, gb, g_ld, table, n_out, n_in, K, Cin, Cout, nbi, ws);
    else if (to == 2 && ti == 2) k_wgrad_bf16<2, 2><<<grid, kWaves * 64, 0, s>>>(xb, x_ld, gb, g_ld, table, n_out, n_in, K, Cin, Cout, nbi, ws);
    else if (to == 2) k_wgrad_bf16<2, 1><<<grid, kWaves * 64, 0, s>>>(xb, x_ld, gb, g_ld, table, n_out, n_in, K, Cin, Cout, nbi, ws);
    else if (ti == 2) k_wgrad_bf16<1, 2><<<grid, kWaves * 64, 0, s>>>(xb, x_ld, gb, g_ld, table, n_out, n_in, K, Cin, Cout, nbi, ws);
    else k_wgrad_bf16<1, 1><<<grid, kWaves * 64, 0, s>>>(xb, x_ld, gb, g_ld, table, n_out, n_in, K, Cin, Cout, nbi, ws);
    }
  } else {
  // [Cout x Cin] is cut into blocks of 32 NBO x 32 NBI; 96 channels take one 3-tile block instead of two 2-tile blocks
  // (a 2 x 2 tiling of 96 x 96 would spend 16 MFMAs where 9 are needed)
  auto tiles = [](int c) { return c <= 32 ? 1 : ((c > 64 && c <= 96) ? 3 : 2); };
  const int to = tiles(Cout), ti = tiles(Cin);
  const int nbo = (int)tl_cdiv(Cout, 32 * to), nbi = (int)tl_cdiv(Cin, 32 * ti);
  const dim3 grid((unsigned)nchunks, (unsigned)K, (unsigned)(nbo * nbi));
#define TL_W(O_, I_)                                                                                                         \
  do {                                                                                                                       \
    if (dtype == TL_BF16) k_wgrad<O_, I_, true><<<grid, kWaves * 64, 0, s>>>(x, x_ld, gout, g_ld, table, n_out, n_in, K, Cin, Cout, nbi, ws); \
    else k_wgrad<O_, I_, false><<<grid, kWaves * 64, 0, s>>>(x, x_ld, gout, g_ld, table, n_out, n_in, K, Cin, Cout, nbi, ws);               \
  } while (0)
  switch (to * 10 + ti) {
    case 11: TL_W(1, 1); break; case 12: TL_W(1, 2); break; case 13: TL_W(1, 3); break;
    case 21: TL_W(2, 1); break; case 22: TL_W(2, 2); break; case 23: TL_W(2, 3); break;
    case 31: TL_W(3, 1); break; case 32: TL_W(3, 2); break; case 33: TL_W(3, 3); break;
  }
#undef TL_W
  }
  const int64_t per = (int64_t)K * Cout * Cin;
  const int64_t nparts = bf16_mfma ? nchunks_used : nchunks * kWaves;
  if (per % 4 == 0 && Cin % 4 == 0 && ((uintptr_t)ws) % 16 == 0 && ((uintptr_t)gw) % 16 == 0) return tl_launch_wgrad_reduce(ws, nparts, per, gw, s, K, Cout, Cin, ref_layout);
  if (ref_layout && K > 1) return TL_ERR_UNSUPPORTED;
  k_wgrad_reduce<<<tl_grid(per, 256), 256, 0, s>>>(ws, nparts, per, gw);
  TL_CHECK_LAUNCH();
  return TL_OK;
}

int tl_conv_wgrad(const void* x, int64_t x_ld, const void* gout, int64_t g_ld, int dtype, const int32_t* table, int64_t n_out, int64_t n_in, int K,
                  int Cin, int Cout, float* gw, float* ws, tl_stream_t stream) {
#ifndef TL_F16_BUILD
  if (dtype == TL_F16) return tl_conv_wgrad_f16(x, x_ld, gout, g_ld, TL_BF16, table, n_out, n_in, K, Cin, Cout, gw, ws, stream);   // the IEEE-half compilation of this unit
#endif
  return wgrad_impl(x, x_ld, gout, g_ld, dtype, table, n_out, n_in, K, Cin, Cout, gw, ws, stream, 0);
}

int tl_conv_wgrad_ref(const void* x, int64_t x_ld, const void* gout, int64_t g_ld, int dtype, const int32_t* table, int64_t n_out, int64_t n_in, int K,
                      int Cin, int Cout, float* gw, float* ws, tl_stream_t stream) {
#ifndef TL_F16_BUILD
  if (dtype == TL_F16) return tl_conv_wgrad_ref_f16(x, x_ld, gout, g_ld, TL_BF16, table, n_out, n_in, K, Cin, Cout, gw, ws, stream);
#endif
  return wgrad_impl(x, x_ld, gout, g_ld, dtype, table, n_out, n_in, K, Cin, Cout, gw, ws, stream, 1);
}

}  // extern "C"
