// Sparse conv, "block-local" form: the 27-tap 32 -> 32 SubM convs of a level whose rows are in the block-local order of tl_blk.hip
// (reference layers: every SubMConv3d of level 1 -- tree_learn/model/blocks.py:55-70 inside `unet.blocks` / `unet.blocks_tail`; 16-bit).
//
// The gather kernels (tl_conv_direct.hip) issue 27 gather slots per output row and sit AT the 32 B/clk/CU roof of the vector-memory path
// with 80 % of the slots fetching nothing.  Here a WAVE owns a unit of <= 64 consecutive rows end to end:
//   stage  : the unit's own rows (contiguous) + its halo rows (tl_blk's list: each distinct outside neighbour ONCE) go global -> LDS by
//            LDS-DMA, 64 B per row, 16-B pieces XOR-swizzled through the source address: ~1.8 staged rows per output row instead of 27 slots;
//   taps   : the A fragments of all 27 taps are ds_read_b128s at the offsets the local rulebook stores (absent neighbour = the stage's
//            all-zero row: no gather slot, no mask), two 32-row tiles per unit share each B fragment; all weights resident in LDS;
//            32x32x16 MFMAs, fp32 accumulators, taps and k-pieces in the order of the direct kernel -> BIT-IDENTICAL results;
//   output : accumulators -> the (dead) stage as an fp32 tile -> row vectors -> residual / up to three views (BatchNorm affine + ReLU),
//            16-B stores.  The NEXT unit's staging is requested as soon as the results have left the stage and BEFORE the stores, so the
//            counted wait at the top of the next unit leaves only the stores outstanding.
// Two waves per SIMD (8 per CU: 55 KB of weights + 8 x 12 KB stages): while one waits for its stage the other computes.  No workgroup
// barrier after the weights are in place.  Rulebook rows and halo indices are requested one and two units ahead with inline-assembly loads
// (fixed position in the instruction stream: the waits are counted).  Units are dealt XCD-contiguously (block b runs on XCD b % 8): the
// halo rows of neighbouring units are hits in that XCD's L2.
#include "tl_conv_internal.h"
#include <atomic>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
static __device__ __forceinline__ u32x4 lds_r128(unsigned a) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); return v; }
#define TL_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory")
#define TL_KEEP(x) asm volatile("" : "+v"(x))
// unit descriptors travel through the scalar cache: hipcc turns a plain uniform load of them into a vector load + s_waitcnt vmcnt(0) +
// readfirstlane (the pointer is not provably read-only), which would drain the staging DMAs and stores at the top of every unit
typedef int i32x4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ i32x4 s_load4(const void* ptr) { i32x4 v; asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(v) : "s"(ptr)); return v; }
#define TL_SWAIT(d) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d) : : "memory")

constexpr int WS_B = 27 * 32 * 64;            // all taps of the weights
constexpr int AFF_B = 4 * 2 * 32 * 4;         // scale / shift of up to three views + of the input (prologue)
constexpr int STAGE_B = 192 * 64;             // positions 0..63 own rows, 64..189 halo rows, 191 the zero row
constexpr int HCH = 8;                        // halo chunks of 16 rows (TL_BLK_HALO_MAX = 126 <= 128)
struct RbRow { u32x4 a, b; uint32_t c; };     // one row of the local rulebook
static __device__ __forceinline__ uint32_t rb_word(const RbRow& r, int w) { return w < 4 ? r.a[w] : (w < 8 ? r.b[w - 4] : r.c); }

// TR: the training-mode epilogue (tl_conv_args.epi_mode: TL_EPI_STATS / TL_EPI_BN_BWD; one view, residual through the shared row-vector
// helper of tl_conv_internal.h): every lane sums its row vectors' summands over all of its wave's units in fp32, the waves' totals
// are combined in fp64 -> one partial row per workgroup, in the format of the other kernel families
// PRO: gather-side prologue relu?(x * in_scale + in_shift) -- the BatchNorm + ReLU that sits in front of every conv of the reference
// (blocks.py:55-70) -- applied ONCE per staged row in LDS (1.8 rows per output row; the gather kernels would have to apply it to 27
// gathered copies, which is why their producers write a second, activated view instead: 118 MB of extra writes per level-1 tensor)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

template <int W, bool RES, int NV, bool TR = false, bool PRO = false>
__global__ void __launch_bounds__(W * 64) k_conv_blk(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, fh = lane >> 5;
  {
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);              // [27][32 cout][4 pieces of 8 cin]
    for (int v = tid; v < 27 * 128; v += W * 64) {
      const int s_ = v & 3, n = (v >> 2) & 31, k = v >> 7;
      *reinterpret_cast<u32x4*>(smem + (k * 32 + n) * 64 + ((s_ ^ ((n >> 2) & 3)) * 16)) = wsrc[v];
    }
    float* aff = reinterpret_cast<float*>(smem + WS_B);                   // [view][scale | shift][32]; training: [mean | rstd | scale | shift][32]
    for (int e = tid; e < 3 * 64; e += W * 64) {
      const int v = e >> 6, c = e & 31, sh = (e >> 5) & 1;
      const float* src = v == 0 ? (sh ? p.out_shift : p.out_scale) : v == 1 ? (sh ? p.out2_shift : p.out2_scale) : (sh ? p.out3_shift : p.out3_scale);
      if constexpr (TR) {
        const float* bsrc[4] = {p.bn_mean, p.bn_rstd, p.bn_scale, p.bn_shift};
        if (e < 128) aff[e] = (p.epi_mode == TL_EPI_BN_BWD) ? bsrc[e >> 5][c] : 0.f;
      } else {
        aff[e] = src ? src[c] : (sh ? 0.f : 1.f);
      }
    }
    if constexpr (PRO) {
      for (int e = tid; e < 64; e += W * 64) aff[192 + e] = (e < 32) ? (p.in_scale ? p.in_scale[e] : 1.f) : (p.in_shift ? p.in_shift[e - 32] : 0.f);
    }
  }
  char* stage = smem + WS_B + AFF_B + wv * STAGE_B;
  for (int e = lane; e < STAGE_B / 16; e += 64) *reinterpret_cast<u32x4*>(stage + e * 16) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem, st_a = lds0 + (unsigned)(WS_B + AFF_B + wv * STAGE_B);
  const unsigned aff_a = lds0 + (unsigned)WS_B + (unsigned)((lane & 3) * 32);   // the lane's 8 channels of a row vector
  unsigned boff[2];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) boff[s_] = lds0 + (unsigned)(fi * 64 + (((2 * s_ + fh) ^ ((fi >> 2) & 3)) * 16));
  const unsigned in_ldb = (unsigned)(p.in_ld * 2);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)((p.n_in - 1) * (int64_t)in_ldb + 64), 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.blk_lrb), 0, (int)(p.n_out * 36), 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.blk_halo), 0, (int)(p.n_out * 128), 0x00020000);
  const unsigned o_ldb[3] = {(unsigned)(p.out_ld * 2), (unsigned)(p.out2_ld * 2), (unsigned)(p.out3_ld * 2)};
  void* const o_ptr[3] = {p.out, p.out2, p.out3};
  __amdgpu_buffer_rsrc_t ro[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) ro[v] = __builtin_amdgcn_make_buffer_rsrc(o_ptr[v], 0, (int)((p.n_out - 1) * (int64_t)o_ldb[v] + 64), 0x00020000);
  const unsigned res_ldb = (unsigned)(p.res_ld * 2);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(RES ? p.res : p.in), 0, RES ? (int)((p.n_out - 1) * (int64_t)res_ldb + 64) : 0, 0x00020000);
  const bool bnb = TR && p.epi_mode == TL_EPI_BN_BWD;
  const bool tr_loads = TR && (bnb || p.res != nullptr);
  const unsigned tr_ldb = (unsigned)((bnb ? p.bn_x_ld : p.res_ld) * 2);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rtr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(tr_loads ? (bnb ? p.bn_x : p.res) : p.in), 0, tr_loads ? (int)((p.n_out - 1) * (int64_t)tr_ldb + 64) : 0, 0x00020000);
  const bool aff_on[3] = {p.out_scale != nullptr, p.out2_scale != nullptr, p.out3_scale != nullptr};
  const bool relu_on[3] = {p.out_relu != 0, p.out2_relu != 0, p.out3_relu != 0};

  // XCD-contiguous deal: XCD x (= blockIdx % 8) owns units [x * q8, (x + 1) * q8), its waves interleave inside that range
  const int nunits = p.blk_counter[0];
  const int q8 = (nunits + 7) >> 3;
  const int xcd = (int)blockIdx.x & 7, wgx = (int)blockIdx.x >> 3;
  const int ulo = xcd * q8, uhi = ulo + q8 < nunits ? ulo + q8 : nunits;
  const int nw = ((int)gridDim.x >> 3) * W;
  const int u0 = ulo + wgx * W + wv;
  const i32x4* units = reinterpret_cast<const i32x4*>(p.blk_unit);
  // (row0, n_own, n_halo, -) of unit u; the load is in flight until TL_SWAIT; a slot past the wave's range reads unit 0 and is voided
  auto desc_req = [&](int u) __attribute__((always_inline)) { return s_load4(units + (u < uhi ? u : 0)); };
  auto desc_fin = [&](i32x4 d, int u) __attribute__((always_inline)) { return u < uhi ? make_int4(d[0], d[1], d[2], 0) : make_int4(0, 0, 0, 0); };
  const unsigned pc16[2] = {(unsigned)(fh * 16), (unsigned)(32 + fh * 16)};

  // d = the unit's descriptor; an exhausted slot (n_own = 0) asks for nothing (out-of-range offsets)
  auto load_hidx = [&](const int4& d, int (&h)[HCH]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < HCH; ++c) {
      const unsigned off = d.y > 0 ? ((unsigned)d.x * 32u + (unsigned)(c * 16 + (lane >> 2))) * 4u : 0xFFFFFFFFu;
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(h[c]) : "v"(off), "s"(rh));
    }
  };
  // the lane's rulebook row: 9 words = 27 ten-bit entries (36 B per row)
  auto load_rb = [&](const int4& d, RbRow (&rb)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const unsigned off = d.y > 0 ? (unsigned)(d.x + t * 32 + fi) * 36u : 0xFFFFFFFFu;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rb[t].a) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(rb[t].b) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:32" : "=v"(rb[t].c) : "v"(off), "s"(rl));
    }
  };
  // staging: 4 own chunks + as many halo chunks as the unit has (the counted wait at the top does not depend on their number)
  auto stage_unit = [&](const int4& d, const int (&h)[HCH]) __attribute__((always_inline)) {
    if (d.y <= 0) return;
    const int row0 = d.x, nown = d.y, nh = d.z;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int pos = c * 16 + (lane >> 2);
      const unsigned off = pos < nown ? (unsigned)(row0 + pos) * in_ldb + (unsigned)((((lane & 3) ^ ((pos >> 2) & 3))) * 16) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + c * 1024), 16, (int)off, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < HCH; ++c) {
      if (c * 16 < nh) {
        const int j = c * 16 + (lane >> 2), pos = 64 + j;
        const unsigned off = j < nh ? (unsigned)h[c] * in_ldb + (unsigned)((((lane & 3) ^ ((pos >> 2) & 3))) * 16) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(stage + 4096 + c * 1024), 16, (int)off, 0, 0, 0);
      }
    }
  };

  int4 dc, d1, d2;
  {
    i32x4 r0 = desc_req(u0), r1 = desc_req(u0 + nw), r2 = desc_req(u0 + 2 * nw);
    TL_SWAIT(r0); TL_SWAIT(r1); TL_SWAIT(r2);
    dc = desc_fin(r0, u0); d1 = desc_fin(r1, u0 + nw); d2 = desc_fin(r2, u0 + 2 * nw);
  }
  int hn[HCH];
  RbRow rbc[2], rbn[2];
  {
    int h0[HCH];
    load_hidx(dc, h0); load_hidx(d1, hn); load_rb(dc, rbc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < HCH; ++c) { TL_KEEP(h0[c]); TL_KEEP(hn[c]); }
    stage_unit(dc, h0);
  }
  [[maybe_unused]] float ts0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ts1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bool firstu = true;
  for (int u = u0; u < uhi; u += nw) {
    const int row0 = dc.x, nown = dc.y;
    i32x4 d3r = desc_req(u + 3 * nw);                                               // waited for behind the taps, used from the next iteration on
    // this unit's stage has landed (requested before the previous unit's stores), and so have its rulebook and the next unit's halo indices
    if (firstu) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NV) : "memory");
    if (!firstu) {
#pragma unroll
      for (int t = 0; t < 2; ++t) { TL_KEEP(rbn[t].a); TL_KEEP(rbn[t].b); TL_KEEP(rbn[t].c); rbc[t] = rbn[t]; }
#pragma unroll
      for (int c = 0; c < HCH; ++c) TL_KEEP(hn[c]);
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) { TL_KEEP(rbc[t].a); TL_KEEP(rbc[t].b); TL_KEEP(rbc[t].c); }
    }
    firstu = false;
    int hcur[HCH];
#pragma unroll
    for (int c = 0; c < HCH; ++c) hcur[c] = hn[c];                                  // indices of unit t + 1 (for the staging after the taps)
    load_hidx(d2, hn);                                                              // unit t + 2
    load_rb(d1, rbn);                                                               // unit t + 1

    if constexpr (PRO) {
      // lane: memory piece m = lane & 3 (channels 8 m ..) of the staged positions (lane >> 2) + 16 i; slot of the piece = m ^ swizzle(pos)
      const int nh_ = dc.z;
      f32x2 isc[4], ish[4];                                                          // channel pairs (2 q, 2 q + 1) of the lane's piece
      {
        const unsigned ia = lds0 + (unsigned)(WS_B + 768 + (lane & 3) * 32);
        const u32x4 s0_ = lds_r128(ia), s1_ = lds_r128(ia + 16), h0_ = lds_r128(ia + 128), h1_ = lds_r128(ia + 144);
        TL_LGKM(0);
        isc[0] = f32x2{__uint_as_float(s0_[0]), __uint_as_float(s0_[1])}; isc[1] = f32x2{__uint_as_float(s0_[2]), __uint_as_float(s0_[3])};
        isc[2] = f32x2{__uint_as_float(s1_[0]), __uint_as_float(s1_[1])}; isc[3] = f32x2{__uint_as_float(s1_[2]), __uint_as_float(s1_[3])};
        ish[0] = f32x2{__uint_as_float(h0_[0]), __uint_as_float(h0_[1])}; ish[1] = f32x2{__uint_as_float(h0_[2]), __uint_as_float(h0_[3])};
        ish[2] = f32x2{__uint_as_float(h1_[0]), __uint_as_float(h1_[1])}; ish[3] = f32x2{__uint_as_float(h1_[2]), __uint_as_float(h1_[3])};
      }
      const int nch = (64 + nh_ + 15) >> 4;                                          // chunks of 16 staged positions (<= 12)
      // ReLU on the rounded pair as a signed 16-bit max (rounding keeps the sign; negative halves and -0 become +0); no ReLU: max with
      // INT16_MIN = identity
      const s16x2 floor2 = p.in_relu ? s16x2{0, 0} : s16x2{(short)-32768, (short)-32768};
      for (int i0 = 0; i0 < nch; i0 += 4) {
        u32x4 dv[4]; unsigned da[4]; bool ok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pos = (lane >> 2) + 16 * (i0 + j);
          ok[j] = (i0 + j) < nch && pos < 64 + nh_ && (pos >= 64 || pos < nown);
          da[j] = st_a + (unsigned)(pos * 64 + (((lane & 3) ^ ((pos >> 2) & 3)) * 16));
          if (!ok[j]) da[j] = st_a + 191u * 64u;                                     // (reads the zero row, writes nothing)
          dv[j] = lds_r128(da[j]);
        }
        TL_LGKM(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          TL_KEEP(dv[j]);
          u32x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 z = __builtin_elementwise_fma(f32x2{bf16_lo(dv[j][q]), bf16_hi(dv[j][q])}, isc[q], ish[q]);
            o[q] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16x2(z[0], z[1])), floor2));
          }
          if (ok[j]) asm volatile("ds_write_b128 %0, %1" ::"v"(da[j]), "v"(o) : "memory");
        }
      }
      TL_LGKM(0);
    }
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    {
      u32x4 A[2][2][2], B[2][2];                                                    // 6 reads per tap
      auto issue = [&](int k, int s_) __attribute__((always_inline)) {
        B[s_][0] = lds_r128(boff[0] + (unsigned)(k * 2048));
        B[s_][1] = lds_r128(boff[1] + (unsigned)(k * 2048));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint32_t wd = rb_word(rbc[t], k / 3);
          const unsigned a0 = st_a + (((wd >> (10 * (k % 3))) & 1023u) << 4);
          A[s_][t][0] = lds_r128(a0 ^ pc16[0]);
          A[s_][t][1] = lds_r128(a0 ^ pc16[1]);
        }
      };
#ifdef TL_DEV
      // developer ablations (tl_set_tuning "dbg"; results are wrong on purpose): 4 = no MFMAs at all, 16 = MFMAs for 5 of the 27 taps only
      // (the level-1 average of PRESENT taps per row: what an ideal present-pairs-only contraction would issue), 32 = also only those 5
      // taps' LDS reads -- the floor of any formulation that skips absent (row, tap) pairs with zero bookkeeping cost
      const int dbg = p.dbg;
      if (dbg & (4 | 16 | 32)) {
        issue(0, 0);
#pragma unroll
        for (int k = 0; k < 27; ++k) {
          const int s_ = k & 1;
          const bool rd = !(dbg & 32) || k + 1 < 5;
          if (k + 1 < 27 && rd) { issue(k + 1, s_ ^ 1); TL_LGKM(6); } else TL_LGKM(0);
          TL_KEEP(B[s_][0]); TL_KEEP(B[s_][1]);
          if (!(dbg & 4) && k < 5) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              TL_KEEP(A[s_][t][0]); TL_KEEP(A[s_][t][1]);
              acc[t] = h16_mfma(A[s_][t][0], B[s_][0], acc[t]);
              acc[t] = h16_mfma(A[s_][t][1], B[s_][1], acc[t]);
            }
          }
        }
      } else
#endif
      {
      issue(0, 0);
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const int s_ = k & 1;
        if (k + 1 < 27) { issue(k + 1, s_ ^ 1); TL_LGKM(6); } else TL_LGKM(0);
        TL_KEEP(B[s_][0]); TL_KEEP(B[s_][1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          TL_KEEP(A[s_][t][0]); TL_KEEP(A[s_][t][1]);
          acc[t] = h16_mfma(A[s_][t][0], B[s_][0], acc[t]);
          acc[t] = h16_mfma(A[s_][t][1], B[s_][1], acc[t]);
        }
      }
      }
    }
    TL_SWAIT(d3r);
    // residual row vectors (lane: row (lane >> 2) + 16 it, channels 8 (lane & 3) ..): requested now, used after the LDS transposition
    [[maybe_unused]] u32x4 rres[4];
    if constexpr (RES) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rr_ = (lane >> 2) + 16 * it;
        const unsigned off = rr_ < nown ? (unsigned)(row0 + rr_) * res_ldb + (unsigned)((lane & 3) * 16) : 0xFFFFFFFFu;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rres[it]) : "v"(off), "s"(rr));
      }
    }
    if constexpr (TR) {                                                              // training: the residual, or the BatchNorm input of TL_EPI_BN_BWD
      if (tr_loads) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int rr_ = (lane >> 2) + 16 * it;
          const unsigned off = rr_ < nown ? (unsigned)(row0 + rr_) * tr_ldb + (unsigned)((lane & 3) * 16) : 0xFFFFFFFFu;
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rres[it]) : "v"(off), "s"(rtr));
        }
      }
    }
    float* ew = reinterpret_cast<float*>(stage);                                    // 64 rows x 36 floats over the dead stage
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) ew[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 36 + fi] = acc[t][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float y[4][8];
    {
      u32x4 e0[4], e1[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rr_ = (lane >> 2) + 16 * it;
        const unsigned ea = st_a + (unsigned)((rr_ * 36 + (lane & 3) * 8) * 4);
        e0[it] = lds_r128(ea); e1[it] = lds_r128(ea + 16);
      }
      TL_LGKM(0);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        TL_KEEP(e0[it]); TL_KEEP(e1[it]);
#pragma unroll
        for (int q = 0; q < 4; ++q) { y[it][q] = __uint_as_float(e0[it][q]); y[it][q + 4] = __uint_as_float(e1[it][q]); }
      }
    }
    if constexpr (RES) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        TL_KEEP(rres[it]);
#pragma unroll
        for (int q = 0; q < 4; ++q) { y[it][2 * q] += bf16_lo(rres[it][q]); y[it][2 * q + 1] += bf16_hi(rres[it][q]); }
      }
    }
    if constexpr (TR) {
      // training (the arithmetic of tl_conv_internal.h's epi_views8_red, so that the stored tensor equals the gather kernels' bit for bit):
      //   TL_EPI_STATS : v = bf16(y + residual);                      summands v, v^2
      //   TL_EPI_BN_BWD: v = [x * scale + shift > 0] ? bf16(y) : 0;    summands v, v * (x - mean) * rstd
      float pm[8], pr[8], ps[8], ph[8];
      if (bnb) {
        const u32x4 m0 = lds_r128(aff_a), m1 = lds_r128(aff_a + 16), r0_ = lds_r128(aff_a + 128), r1_ = lds_r128(aff_a + 144);
        const u32x4 s0_ = lds_r128(aff_a + 256), s1_ = lds_r128(aff_a + 272), h0_ = lds_r128(aff_a + 384), h1_ = lds_r128(aff_a + 400);
        TL_LGKM(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          pm[q] = __uint_as_float(m0[q]); pm[q + 4] = __uint_as_float(m1[q]); pr[q] = __uint_as_float(r0_[q]); pr[q + 4] = __uint_as_float(r1_[q]);
          ps[q] = __uint_as_float(s0_[q]); ps[q + 4] = __uint_as_float(s1_[q]); ph[q] = __uint_as_float(h0_[q]); ph[q + 4] = __uint_as_float(h1_[q]);
        }
      }
      if (tr_loads) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      u32x4 o[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rr_ = (lane >> 2) + 16 * it;
        float xv[8];
        if (tr_loads) {
          TL_KEEP(rres[it]);
#pragma unroll
          for (int q = 0; q < 4; ++q) { xv[2 * q] = bf16_lo(rres[it][q]); xv[2 * q + 1] = bf16_hi(rres[it][q]); }
        }
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (bnb) {
            const bool keep = !p.bn_relu || fmaf(xv[q], ps[q], ph[q]) > 0.f;
            v[q] = keep ? round_to(y[it][q], true) : 0.f;
          } else {
            v[q] = round_to(tr_loads ? y[it][q] + xv[q] : y[it][q], true);
          }
        }
        if (rr_ < nown) {
#pragma unroll
          for (int q = 0; q < 8; ++q) { ts0[q] += v[q]; ts1[q] += bnb ? v[q] * ((xv[q] - pm[q]) * pr[q]) : v[q] * v[q]; }
        }
        o[it] = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
      }
      stage_unit(d1, hcur);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rr_ = (lane >> 2) + 16 * it;
        const unsigned off = rr_ < nown ? (unsigned)(row0 + rr_) * o_ldb[0] + (unsigned)((lane & 3) * 16) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o[it]), ro[0], (int)off, 0, 0);
      }
    } else {
    // the stage is free (the results are in registers): the next unit's staging goes out BEFORE this unit's stores
    stage_unit(d1, hcur);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float sc[8], sh[8];
      if (aff_on[v]) {
        const u32x4 s0 = lds_r128(aff_a + (unsigned)(v * 256)), s1 = lds_r128(aff_a + (unsigned)(v * 256 + 16));
        const u32x4 h0 = lds_r128(aff_a + (unsigned)(v * 256 + 128)), h1 = lds_r128(aff_a + (unsigned)(v * 256 + 144));
        TL_LGKM(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sc[q] = __uint_as_float(s0[q]); sc[q + 4] = __uint_as_float(s1[q]);
          sh[q] = __uint_as_float(h0[q]); sh[q + 4] = __uint_as_float(h1[q]);
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float z[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) z[q] = y[it][q];
        if (aff_on[v]) {
#pragma unroll
          for (int q = 0; q < 8; ++q) z[q] = fmaf(z[q], sc[q], sh[q]);
        }
        if (relu_on[v]) {
#pragma unroll
          for (int q = 0; q < 8; ++q) z[q] = fmaxf(z[q], 0.f);
        }
        const u32x4 o = {pack_bf16x2(z[0], z[1]), pack_bf16x2(z[2], z[3]), pack_bf16x2(z[4], z[5]), pack_bf16x2(z[6], z[7])};
        const int rr_ = (lane >> 2) + 16 * it;
        const unsigned off = rr_ < nown ? (unsigned)(row0 + rr_) * o_ldb[v] + (unsigned)((lane & 3) * 16) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), ro[v], (int)off, 0, 0);
      }
    }
    }
    dc = d1; d1 = d2; d2 = desc_fin(d3r, u + 3 * nw);
  }
  if constexpr (TR) {
    // lanes l, l + 4, ... hold the same eight channels: add them (fp64), then the waves in wave order -> the workgroup's partial row
    double* Ds = reinterpret_cast<double*>(smem + WS_B + AFF_B + W * STAGE_B);           // [W][2][32]
    double d0[8], d1_[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { d0[q] = (double)ts0[q]; d1_[q] = (double)ts1[q]; }
#pragma unroll
    for (int off = 4; off < 64; off <<= 1)
#pragma unroll
      for (int q = 0; q < 8; ++q) { d0[q] += __shfl_xor(d0[q], off, 64); d1_[q] += __shfl_xor(d1_[q], off, 64); }
    if (lane < 4) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { Ds[(wv * 2 + 0) * 32 + lane * 8 + q] = d0[q]; Ds[(wv * 2 + 1) * 32 + lane * 8 + q] = d1_[q]; }
    }
    __syncthreads();
    if (tid < 64) {
      double t = 0.0;
      for (int w = 0; w < W; ++w) t += Ds[(w * 2 + (tid >> 5)) * 32 + (tid & 31)];
      p.red_part[(int64_t)blockIdx.x * 64 + tid] = t;
    }
  }
}

template <int W, bool RES, int NV, bool TR = false, bool PRO = false>
int launch_blk(const ConvP& p, hipStream_t s) {
  constexpr size_t lds = (size_t)WS_B + AFF_B + (size_t)W * STAGE_B + (TR ? (size_t)W * 2 * 32 * 8 : 0);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static TlAttrOnce attr_once;                     // per kernel instantiation AND device (the attribute is per device)
  if (!tl_lds_attr(attr_once, reinterpret_cast<const void*>(&k_conv_blk<W, RES, NV, TR, PRO>), 160 * 1024)) return TL_ERR_LAUNCH;
  k_conv_blk<W, RES, NV, TR, PRO><<<256, W * 64, lds, s>>>(p);
  if (TR && p.red_nparts) *p.red_nparts = 256;
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

}  // namespace

// 27 taps, 32 -> 32 channels, 16-bit, rows in the block-local order with the staged rulebook of tl_blk_build; no gather-side prologue, no
// training epilogue.  Every view 16-B aligned with a row stride that is a multiple of 8 elements; buffers below 4 GB.
int tl_launch_conv_blk(const ConvP& p, hipStream_t s) {
  if (!p.blk_unit || !p.blk_counter || !p.blk_halo || !p.blk_lrb) return TL_ERR_UNSUPPORTED;
  if (p.K != 27 || p.Cin != 32 || p.Cout != 32 || p.n_in != p.n_out) return TL_ERR_UNSUPPORTED;
  const bool pro = p.in_scale || p.in_relu;
  if (p.n_out >= (1 << 25)) return TL_ERR_UNSUPPORTED;
  auto big = [&](int64_t ld) { return (p.n_out - 1) * ld * 2 + 64 >= 0x7FFFFFFFll * 2; };
  if (big(p.in_ld) || big(p.out_ld) || (p.out2 && big(p.out2_ld)) || (p.out3 && big(p.out3_ld)) || (p.res && big(p.res_ld))) return TL_ERR_UNSUPPORTED;
  if (p.out3 && !p.out2) return TL_ERR_UNSUPPORTED;
  if (p.epi_mode != TL_EPI_NONE) {
    if (p.out2 || p.out_scale || p.out_relu || p.out_ld % 8 || ((uintptr_t)p.out) % 16) return TL_ERR_UNSUPPORTED;
    if (pro && p.epi_mode != TL_EPI_STATS) return TL_ERR_UNSUPPORTED;
    // (a residual goes through the shared row stage; with the prologue the training forward applies its BatchNorm + ReLU at staging, so the
    //  activated tensor is never written -- the weight gradient recomputes it on its own stream)
    return pro ? launch_blk<8, false, 1, true, true>(p, s) : launch_blk<8, false, 1, true>(p, s);
  }
  const int nv = p.out3 ? 3 : p.out2 ? 2 : 1;
  if (pro) {                                                       // prologue form: one view (what the engine's level-1 first convs need)
    if (nv != 1) return TL_ERR_UNSUPPORTED;
    return p.res ? launch_blk<8, true, 1, false, true>(p, s) : launch_blk<8, false, 1, false, true>(p, s);
  }
  if (p.res) {
    switch (nv) {
      case 1: return launch_blk<8, true, 1>(p, s);
      case 2: return launch_blk<8, true, 2>(p, s);
      default: return launch_blk<8, true, 3>(p, s);
    }
  }
  switch (nv) {
    case 1: return launch_blk<8, false, 1>(p, s);
    case 2: return launch_blk<8, false, 2>(p, s);
    default: return launch_blk<8, false, 3>(p, s);
  }
}
