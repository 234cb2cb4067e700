// Sparse conv, "block-local" form for the parity-fast mode (compute_dtype "bf16x3": fp32 storage, split-bf16 contraction): the 27-tap
// 32 -> 32 SubM convs of level 1 on rows in the block-local order of tl_blk.hip (reference layers: every SubMConv3d of level 1 --
// tree_learn/model/blocks.py:55-70 inside `unet.blocks` / `unet.blocks_tail`; the reference's own inference arithmetic is fp32,
// tree_learn/util/pipeline.py:86).
//
// The gather kernels serve this mode by fetching every fp32 neighbour row 27 times and splitting it into its bf16 head and tail once PER
// TAP (x3_split8: ~40 VALU instructions per 16 channels and tap); level 1 took 5.2 of the mode's 23 ms that way (0.64 ms per conv against
// 0.115 for the bf16 staged-unit kernel).  Here the staged-unit design of tl_conv_blk.hip carries over with ONE change of granularity: a
// launch contracts SIXTEEN input channels, so that a staged position is still 64 B and everything that kernel's LDS budget rests on holds:
//   stage  : 64 B per position = 16 fp32 channels of a row (own rows contiguous, each distinct halo row once: tl_blk's lists), by LDS-DMA;
//   split  : ONE pass over the staged positions turns the 16 floats -- after the optional BatchNorm + ReLU prologue, applied in fp32 --
//            into hi | lo bf16 halves IN PLACE (hi = upper 16 bits, lo = bf16(x - hi): tl_conv_internal.h), 16-B slots 0 / 1 = the hi
//            operand of lane half 0 / 1, slots 2 / 3 = the lo operand: 1.8 splits per output row instead of 27;
//   taps   : per tap and 32-row tile two ds_read_b128 (hi, lo) through the local rulebook and three 32x32x16 MFMAs (lo.Whi + hi.Wlo +
//            hi.Whi) against the tap's split weights, all 27 taps' hi | lo weights of the 16 channels resident in LDS (54 KB, the
//            tl_pack_weight_x3 layout read slot-wise);
//   output : fp32 rows (two 16-B stores per 8 channels); the launch of the SECOND channel slice adds the first slice's raw sums (read back
//            from `out`: a wave re-reads only rows it is about to overwrite), the residual, and writes the requested views.
// A 32 -> 32 conv is therefore two launches (tl_launch_conv_blk_x3 issues both); fp32 sums of 27 x 16 x 3 products per launch, the two
// launches' sums added in fp32 -- another summation order than the gather kernels' (per tap: 32 channels), same products.
// Eight waves per CU (54 KB weights + 1 KB affines + 8 x 12 KB stages), no workgroup barrier after the weights are in place, counted waits.
#include "tl_conv_internal.h"
#include <atomic>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
static __device__ __forceinline__ u32x4 lds_r128(unsigned a) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); return v; }
#define TL_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory")
#define TL_KEEP(x) asm volatile("" : "+v"(x))
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ i32x4 s_load4(const void* ptr) { i32x4 v; asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(v) : "s"(ptr)); return v; }
#define TL_SWAIT(d) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d) : : "memory")

constexpr int WS_B = 27 * 32 * 64;            // all taps: [27][32 cout][hi fh0 | hi fh1 | lo fh0 | lo fh1] x 16 B
constexpr int AFF_B = 4 * 2 * 32 * 4;         // scale / shift of up to three views (32 channels each) + of the 16 input channels
constexpr int STAGE_B = 192 * 64;             // positions 0..63 own rows, 64..189 halo rows, 191 the zero row
constexpr int HCH = 8;
struct RbRow { u32x4 a, b; uint32_t c; };
static __device__ __forceinline__ uint32_t rb_word(const RbRow& r, int w) { return w < 4 ? r.a[w] : (w < 8 ? r.b[w - 4] : r.c); }

struct X3P {                                   // what differs between the two launches of one conv
  int slice;                                   // 16-channel slice of the 32 input channels this launch contracts (weights' J)
  int part;                                    // != 0: add the sums already in `out` (the previous slice's launch)
  int chunk, nchunks;                          // this launch serves chunk `chunk` of `nchunks` of every XCD's range of units
};

// RES: fp32 residual; NV: output views; the last launch of a conv carries RES / NV / affines, the first one NV = 1 raw.
template <int W, bool RES, int NV>
__global__ void __launch_bounds__(W * 64) k_conv_blk_x3(ConvP p, X3P xp) {
  extern __shared__ __attribute__((aligned(64))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, fh = lane >> 5;
  const bool PART = xp.part != 0;
  {
    // tl_pack_weight_x3 record of (tap k, cout n), Cin = 32: 128 B = hi pieces at 16-B slots 2 J + fh, lo pieces at 4 + 2 J + fh
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w_x3);
    const int J = xp.slice;
    for (int v = tid; v < 27 * 128; v += W * 64) {
      const int s_ = v & 3, n = (v >> 2) & 31, k = v >> 7;
      const int src_slot = (s_ < 2 ? 2 * J + s_ : 4 + 2 * J + (s_ - 2));
      *reinterpret_cast<u32x4*>(smem + (k * 32 + n) * 64 + ((s_ ^ ((n >> 2) & 3)) * 16)) = wsrc[(k * 32 + n) * 8 + src_slot];
    }
    float* aff = reinterpret_cast<float*>(smem + WS_B);                   // [view][scale | shift][32]; input: [scale 16 | shift 16] at float 192
    for (int e = tid; e < 3 * 64; e += W * 64) {
      const int v = e >> 6, c = e & 31, sh = (e >> 5) & 1;
      const float* src = v == 0 ? (sh ? p.out_shift : p.out_scale) : v == 1 ? (sh ? p.out2_shift : p.out2_scale) : (sh ? p.out3_shift : p.out3_scale);
      aff[e] = src ? src[c] : (sh ? 0.f : 1.f);
    }
    for (int e = tid; e < 32; e += W * 64) aff[192 + e] = (e < 16) ? (p.in_scale ? p.in_scale[e] : 1.f) : (p.in_shift ? p.in_shift[e - 16] : 0.f);
  }
  char* stage = smem + WS_B + AFF_B + wv * STAGE_B;
  for (int e = lane; e < STAGE_B / 16; e += 64) *reinterpret_cast<u32x4*>(stage + e * 16) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem, st_a = lds0 + (unsigned)(WS_B + AFF_B + wv * STAGE_B);
  const unsigned aff_a = lds0 + (unsigned)WS_B + (unsigned)((lane & 3) * 32);   // the lane's 8 channels of a row vector
  unsigned boff[2];                                                              // hi / lo weight fragment of the lane
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) boff[s_] = lds0 + (unsigned)(fi * 64 + (((2 * s_ + fh) ^ ((fi >> 2) & 3)) * 16));
  const unsigned in_ldb = (unsigned)(p.in_ld * 4);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)((p.n_in - 1) * (int64_t)in_ldb + 64), 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.blk_lrb), 0, (int)(p.n_out * 36), 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.blk_halo), 0, (int)(p.n_out * 128), 0x00020000);
  const unsigned o_ldb[3] = {(unsigned)(p.out_ld * 4), (unsigned)(p.out2_ld * 4), (unsigned)(p.out3_ld * 4)};
  void* const o_ptr[3] = {p.out, p.out2, p.out3};
  __amdgpu_buffer_rsrc_t ro[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) ro[v] = __builtin_amdgcn_make_buffer_rsrc(o_ptr[v], 0, (int)((p.n_out - 1) * (int64_t)o_ldb[v] + 128), 0x00020000);
  const unsigned res_ldb = (unsigned)(p.res_ld * 4);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(RES ? p.res : p.in), 0, RES ? (int)((p.n_out - 1) * (int64_t)res_ldb + 128) : 0, 0x00020000);
  const bool aff_on[3] = {p.out_scale != nullptr, p.out2_scale != nullptr, p.out3_scale != nullptr};
  const bool relu_on[3] = {p.out_relu != 0, p.out2_relu != 0, p.out3_relu != 0};

  // XCD-contiguous deal of the units (tl_conv_blk.hip)
  const int nunits = p.blk_counter[0];
  const int q8 = (nunits + 7) >> 3;
  const int xcd = (int)blockIdx.x & 7, wgx = (int)blockIdx.x >> 3;
  const int xlo = xcd * q8, xhi = xlo + q8 < nunits ? xlo + q8 : nunits;
  const int qc = (q8 + xp.nchunks - 1) / xp.nchunks;
  const int ulo = xlo + xp.chunk * qc, uhi = ulo + qc < xhi ? ulo + qc : xhi;
  const int nw = ((int)gridDim.x >> 3) * W;
  const int u0 = ulo + wgx * W + wv;
  const i32x4* units = reinterpret_cast<const i32x4*>(p.blk_unit);
  auto desc_req = [&](int u) __attribute__((always_inline)) { return s_load4(units + (u < uhi ? u : 0)); };
  auto desc_fin = [&](i32x4 d, int u) __attribute__((always_inline)) { return u < uhi ? make_int4(d[0], d[1], d[2], 0) : make_int4(0, 0, 0, 0); };
  const unsigned pc16[2] = {(unsigned)(fh * 16), (unsigned)(32 + fh * 16)};     // hi / lo operand slot of the lane half (before the position's swizzle)

  auto load_hidx = [&](const int4& d, int (&h)[HCH]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < HCH; ++c) {
      const unsigned off = d.y > 0 ? ((unsigned)d.x * 32u + (unsigned)(c * 16 + (lane >> 2))) * 4u : 0xFFFFFFFFu;
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(h[c]) : "v"(off), "s"(rh));
    }
  };
  auto load_rb = [&](const int4& d, RbRow (&rb)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const unsigned off = d.y > 0 ? (unsigned)(d.x + t * 32 + fi) * 36u : 0xFFFFFFFFu;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rb[t].a) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(rb[t].b) : "v"(off), "s"(rl));
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:32" : "=v"(rb[t].c) : "v"(off), "s"(rl));
    }
  };
  // staging: the 64 B of the launch's channel slice of every own / halo row; 16-B piece m of a position lands in slot m ^ swizzle(pos)
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
  // the lane's four input channels (piece m = lane & 3 of the slice): BatchNorm scale / shift of the prologue (1 / 0 when there is none)
  float isc[4], ish[4];
  {
    const unsigned ia = lds0 + (unsigned)(WS_B + 768 + (lane & 3) * 16);
    const u32x4 s_ = lds_r128(ia), h_ = lds_r128(ia + 64);
    TL_LGKM(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { isc[q] = __uint_as_float(s_[q]); ish[q] = __uint_as_float(h_[q]); }
  }
  const float in_floor = p.in_relu ? 0.f : -__builtin_huge_valf();
  bool firstu = true;
  for (int u = u0; u < uhi; u += nw) {
    const int row0 = dc.x, nown = dc.y;
    i32x4 d3r = desc_req(u + 3 * nw);
    // this unit's stage has landed (requested before the previous unit's stores), and so have its rulebook and the next unit's halo indices
    if (firstu) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * NV) : "memory");
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
    for (int c = 0; c < HCH; ++c) hcur[c] = hn[c];
    load_hidx(d2, hn);
    load_rb(d1, rbn);

    {
      // the split pass: lane = memory piece m = lane & 3 (channels 4 m .. 4 m + 3 of the slice) of the staged positions (lane >> 2) + 16 i.
      // A chunk of 16 positions is read completely (every lane its piece), waited for, then rewritten: hi halves of piece m -> 8 B at
      // slot (m & 1), half (m >> 1); lo halves -> slot 2 + (m & 1), half (m >> 1)   (slots swizzled by the position like the pieces were)
      const int nh_ = dc.z;
      const int nch = (64 + nh_ + 15) >> 4;
      const int m = lane & 3;
      for (int i0 = 0; i0 < nch; i0 += 4) {
        u32x4 dv[4]; unsigned pb[4]; bool ok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pos = (lane >> 2) + 16 * (i0 + j);
          ok[j] = (i0 + j) < nch && pos < 64 + nh_ && (pos >= 64 || pos < nown);
          const int sw = (pos >> 2) & 3;
          pb[j] = st_a + (unsigned)(pos * 64);
          unsigned da = pb[j] + (unsigned)((m ^ sw) * 16);
          if (!ok[j]) da = st_a + 191u * 64u;                                      // (reads the zero row, writes nothing)
          dv[j] = lds_r128(da);
        }
        TL_LGKM(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          TL_KEEP(dv[j]);
          const int pos = (lane >> 2) + 16 * (i0 + j);
          const int sw = (pos >> 2) & 3;
          uint32_t xb[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) xb[q] = __float_as_uint(fmaxf(fmaf(__uint_as_float(dv[j][q]), isc[q], ish[q]), in_floor));
          u32x2 hi, lo;
          hi[0] = __builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u);
          hi[1] = __builtin_amdgcn_perm(xb[3], xb[2], 0x07060302u);
          lo[0] = x3_pack2(__uint_as_float(xb[0]) - __uint_as_float(xb[0] & 0xFFFF0000u), __uint_as_float(xb[1]) - __uint_as_float(xb[1] & 0xFFFF0000u));
          lo[1] = x3_pack2(__uint_as_float(xb[2]) - __uint_as_float(xb[2] & 0xFFFF0000u), __uint_as_float(xb[3]) - __uint_as_float(xb[3] & 0xFFFF0000u));
          const unsigned ah = pb[j] + (unsigned)((((m & 1)) ^ sw) * 16 + (m >> 1) * 8);
          const unsigned al = pb[j] + (unsigned)(((2 + (m & 1)) ^ sw) * 16 + (m >> 1) * 8);
          if (ok[j]) {
            asm volatile("ds_write_b64 %0, %1" ::"v"(ah), "v"(hi) : "memory");
            asm volatile("ds_write_b64 %0, %1" ::"v"(al), "v"(lo) : "memory");
          }
        }
        TL_LGKM(0);                                                                 // (the next chunk's reads never touch these positions, but keep the queue short)
      }
    }
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    {
      u32x4 A[2][2][2], B[2][2];                                                    // [buffer][tile][hi | lo], [buffer][hi | lo]: 6 reads per tap
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
      issue(0, 0);
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const int s_ = k & 1;
        if (k + 1 < 27) { issue(k + 1, s_ ^ 1); TL_LGKM(6); } else TL_LGKM(0);
        TL_KEEP(B[s_][0]); TL_KEEP(B[s_][1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          TL_KEEP(A[s_][t][0]); TL_KEEP(A[s_][t][1]);
          mma16_x3(acc[t], A[s_][t][0], A[s_][t][1], B[s_][0], B[s_][1]);
        }
      }
    }
    TL_SWAIT(d3r);
    float* ew = reinterpret_cast<float*>(stage);                                    // 64 rows x 36 floats over the dead stage
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) ew[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 36 + fi] = acc[t][r];
    // row vectors of the previous slice's sums and of the residual (lane: row (lane >> 2) + 16 it, channels 8 (lane & 3) ..): requested once the
    // accumulators have left the registers, used after the LDS transposition
    [[maybe_unused]] u32x4 rpart[4][2], rres[4][2];
    if (PART) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rr_ = (lane >> 2) + 16 * it;
        const unsigned off = rr_ < nown ? (unsigned)(row0 + rr_) * o_ldb[0] + (unsigned)((lane & 3) * 32) : 0xFFFFFFFFu;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rpart[it][0]) : "v"(off), "s"(ro[0]));
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(rpart[it][1]) : "v"(off), "s"(ro[0]));
      }
    }
    if constexpr (RES) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rr_ = (lane >> 2) + 16 * it;
        const unsigned off = rr_ < nown ? (unsigned)(row0 + rr_) * res_ldb + (unsigned)((lane & 3) * 32) : 0xFFFFFFFFu;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rres[it][0]) : "v"(off), "s"(rr));
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(rres[it][1]) : "v"(off), "s"(rr));
      }
    }
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
    if (PART || RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (PART) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        TL_KEEP(rpart[it][0]); TL_KEEP(rpart[it][1]);
#pragma unroll
        for (int q = 0; q < 4; ++q) { y[it][q] += __uint_as_float(rpart[it][0][q]); y[it][q + 4] += __uint_as_float(rpart[it][1][q]); }
      }
    }
    if constexpr (RES) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        TL_KEEP(rres[it][0]); TL_KEEP(rres[it][1]);
#pragma unroll
        for (int q = 0; q < 4; ++q) { y[it][q] += __uint_as_float(rres[it][0][q]); y[it][q + 4] += __uint_as_float(rres[it][1][q]); }
      }
    }
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
        typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned uv4;
        const uv4 o0 = {__float_as_uint(z[0]), __float_as_uint(z[1]), __float_as_uint(z[2]), __float_as_uint(z[3])};
        const uv4 o1 = {__float_as_uint(z[4]), __float_as_uint(z[5]), __float_as_uint(z[6]), __float_as_uint(z[7])};
        const int rr_ = (lane >> 2) + 16 * it;
        const unsigned off = rr_ < nown ? (unsigned)(row0 + rr_) * o_ldb[v] + (unsigned)((lane & 3) * 32) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b128(o0, ro[v], (int)off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o1, ro[v], (int)(off == 0xFFFFFFFFu ? off : off + 16u), 0, 0);
      }
    }
    dc = d1; d1 = d2; d2 = desc_fin(d3r, u + 3 * nw);
  }
}

int g_x3_chunks = 0;          // tl_set_tuning("x3_chunks"): 0 = one chunk (default: chunking measured slower, profiles/r6_x3/x3_blk.txt)

template <int W, bool RES, int NV>
int launch_x3(const ConvP& p, const X3P& xp, hipStream_t s) {
  constexpr size_t lds = (size_t)WS_B + AFF_B + (size_t)W * STAGE_B;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static TlAttrOnce attr_once;                     // per kernel instantiation AND device (the attribute is per device)
  if (!tl_lds_attr(attr_once, reinterpret_cast<const void*>(&k_conv_blk_x3<W, RES, NV>), 160 * 1024)) return TL_ERR_LAUNCH;
  k_conv_blk_x3<W, RES, NV><<<256, W * 64, lds, s>>>(p, xp);
  return hipGetLastError() == hipSuccess ? TL_OK : TL_ERR_LAUNCH;
}

}  // namespace

// 27 taps, 32 -> 32 channels, fp32 rows in the block-local order with the staged rulebook of tl_blk_build, split-bf16 weights (p.w_x3);
// optional gather-side prologue (in_scale / in_shift / in_relu), fp32 residual, up to two views.  Issues the conv's two launches.
int tl_launch_conv_blk_x3(const ConvP& p, hipStream_t s) {
  if (!p.blk_unit || !p.blk_counter || !p.blk_halo || !p.blk_lrb || !p.w_x3) return TL_ERR_UNSUPPORTED;
  if (p.K != 27 || p.Cin != 32 || p.Cout != 32 || p.n_in != p.n_out || p.epi_mode != TL_EPI_NONE || p.out3) return TL_ERR_UNSUPPORTED;
  if (p.n_out >= (1 << 25)) return TL_ERR_UNSUPPORTED;
  if (p.in_relu && !p.in_scale) return TL_ERR_UNSUPPORTED;
  auto big = [&](int64_t ld) { return (p.n_out - 1) * ld * 4 + 128 >= 0x7FFFFFFFll * 2; };
  if (big(p.in_ld) || big(p.out_ld) || (p.out2 && big(p.out2_ld)) || (p.res && big(p.res_ld))) return TL_ERR_UNSUPPORTED;
  auto al16 = [](const void* q, int64_t ld) { return ((uintptr_t)q) % 16 == 0 && ld % 4 == 0; };
  if (!al16(p.in, p.in_ld) || !al16(p.out, p.out_ld) || (p.out2 && !al16(p.out2, p.out2_ld)) || (p.res && !al16(p.res, p.res_ld)) || ((uintptr_t)p.w_x3) % 16) return TL_ERR_UNSUPPORTED;
  // launch A: input channels 0..15 -> raw sums in `out`;  launch B: input channels 16..31, + the sums of launch A, + the residual, -> the
  // requested views.  Optionally (tl_set_tuning "x3_chunks" > 0) the rows are served in chunks, A and B of a chunk back to back, so that B
  // would find A's sums (128 B per row) in the memory-side cache -- measured on the config-2 level (237 MB of sums): 0.354 ms unchunked,
  // 0.362 / 0.368 / 0.398 with 2 / 3 / 4 chunks (every launch stages its 54 KB of weights per workgroup and drains 256 workgroups): off.
  ConvP a = p;
  a.res = nullptr; a.res_ld = 0; a.out_scale = a.out_shift = nullptr; a.out_relu = 0; a.out2 = nullptr; a.out2_scale = a.out2_shift = nullptr; a.out2_relu = 0;
  ConvP b = p;
  b.in = static_cast<const float*>(p.in) + 16;
  if (p.in_scale) { b.in_scale = p.in_scale + 16; b.in_shift = p.in_shift + 16; }
  int nchunks = 1;
  if (g_x3_chunks > 0) {                                      // at least g_x3_chunks, and no chunk's sums above ~120 MB (the memory-side cache holds 256 MB)
    const int64_t by_size = (p.n_out * 128 + (120ll << 20) - 1) / (120ll << 20);
    nchunks = (int)(by_size > g_x3_chunks ? by_size : g_x3_chunks);
    if (nchunks > 32) nchunks = 32;
    while (nchunks > 1 && p.n_out / nchunks < 400000) --nchunks;
  }
  for (int c = 0; c < nchunks; ++c) {
    int rc = launch_x3<8, false, 1>(a, X3P{0, 0, c, nchunks}, s);
    if (rc != TL_OK) return rc;
    const X3P xb{1, 1, c, nchunks};
    if (p.res) rc = p.out2 ? launch_x3<8, true, 2>(b, xb, s) : launch_x3<8, true, 1>(b, xb, s);
    else rc = p.out2 ? launch_x3<8, false, 2>(b, xb, s) : launch_x3<8, false, 1>(b, xb, s);
    if (rc != TL_OK) return rc;
  }
  return TL_OK;
}

int tl_conv_blk_x3_set_chunks(int n) { g_x3_chunks = n; return TL_OK; }
