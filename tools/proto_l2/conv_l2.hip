// PROTOTYPE (developer tool, not part of the library): the staged-unit form of the level-2 64 -> 64 SubM conv that DESIGN.md R4.9 describes.
//
// A workgroup of 4 waves (one per SIMD, 96 rows each) owns a unit of 384 consecutive rows of the block-local order of level 2 (run_l2.py builds the order, the halo lists
// and the ten-bit local rulebook with numpy).  Whole 128-B rows -- own rows + <= 639 halo rows -- are staged ONCE per unit into a 1 024-row
// LDS stage (128 KB, one workgroup per CU); the optional BatchNorm + ReLU of the consumer is applied on the way in (registers, no second
// view needed).  Every wave then contracts its 96 rows x 64 output channels (six 32 x 32 accumulator tiles): per tap 12 A fragments from
// the stage (addressed through the rulebook entry) and 8 B fragments read as fragment-order vectors straight from L1 / L2 (all four waves
// of the workgroup read the same 8 KB of the tap) feed 24 MFMAs.  No barrier and no weight staging inside the tap loop.
//
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/proto_l2/conv_l2.hip -o tools/proto_l2/libproto_l2.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;

namespace {

constexpr int U = 384, WAVES = 4, NT = WAVES * 64, MT = U / WAVES / 32;      // rows of a unit, waves per workgroup, 32-row tiles per wave (3)
constexpr int POS = 1024, ZERO = 1023;                   // stage positions; the row absent taps read
constexpr int HMAX = POS - 1 - U;                        // halo rows a unit may have (639)

struct P {
  const uint16_t* x; const uint16_t* wfrag; uint16_t* out;
  const int32_t* halo;       // [units][HMAX] new-row ids, ascending
  const int32_t* nhalo;      // [units]
  const uint32_t* lrb;       // [n][9]: 27 ten-bit stage positions per row
  const float* in_scale; const float* in_shift;   // optional staging prologue (both or none): relu(x * scale + shift)
  int64_t n; int units;
  int mode;                  // timing ablations: 1 no staging, 2 no tap loop, 4 no epilogue
};

__device__ __forceinline__ u32x4 lds_r128(unsigned a) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); return v; }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { const bf16x2 v = {(__bf16)lo, (__bf16)hi}; return __builtin_bit_cast(uint32_t, v); }

// byte offset of the 16-B piece `pc` (0..7) of stage position `pos`: pieces XOR-swizzled by the position so that lanes reading
// different rows spread over the banks
__device__ __forceinline__ unsigned st_off(unsigned pos, unsigned pc) { return pos * 128u + ((pc ^ (pos & 7u)) << 4); }

template <int TAPS>      // 27; 1 = the timing ablation "everything but the tap loop"
__global__ void __launch_bounds__(NT) k_conv_l2(P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];            // [POS][128 B]
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  // the weights through a buffer resource: lane offset in a register, the (tap, block, step) offset as the scalar operand -- with plain
  // pointers hipcc kept a 64-bit address pair per unrolled load alive across the unit loop (384 spilled registers)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wfrag), 0, 27 * 64 * 64 * 2, 0x00020000);
  const bool pro = p.in_scale != nullptr;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, (int)(p.n * 128), 0x00020000);
  for (int e = tid; e < 8; e += NT) *reinterpret_cast<u32x4*>(smem + ZERO * 128 + e * 16) = u32x4{0u, 0u, 0u, 0u};

  for (int u = blockIdx.x; u < p.units; u += gridDim.x) {
    const int64_t row0 = (int64_t)u * U;
    const int nown = (int)(p.n - row0 < U ? p.n - row0 : U);
    const int nh = p.nhalo[u];
    __syncthreads();                                                     // the previous unit's epilogue is done with the stage
    // ---- staging by LDS-DMA: one instruction = 8 consecutive stage positions (1 KB: lane l -> row l >> 3, slot l & 7); the slot holds piece
    // slot ^ (pos & 7), so the swizzle is applied on the source side.  A wave first requests the halo row ids of all its instructions, then
    // issues the DMAs back to back (nothing passes through registers, everything is in flight at once).
    const int npos = U + nh;
    const int nq = (npos + 7) >> 3;
    constexpr int QMAX = (POS / 8 + WAVES - 1) / WAVES;                  // instructions per wave (22)
    int hrow[QMAX];
#pragma unroll
    for (int i = 0; i < QMAX; ++i) {
      const int q = wv + i * WAVES, pos = q * 8 + (lane >> 3);
      hrow[i] = (q < nq && pos >= U && pos < npos) ? p.halo[(int64_t)u * HMAX + pos - U] : -1;
    }
#pragma unroll
    for (int i = 0; i < QMAX; ++i) {
      const int q = wv + i * WAVES, pos = q * 8 + (lane >> 3);
      if (q < nq && !(p.mode & 1)) {                                     // wave-uniform
        const int64_t r = pos < U ? (pos < nown ? row0 + pos : -1) : (int64_t)hrow[i];
        const unsigned off = r >= 0 ? (unsigned)r * 128u + (unsigned)((((lane & 7) ^ (pos & 7))) * 16) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + q * 1024), 16, (int)off, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pro) {                                                           // BatchNorm + ReLU of the consumer on the staged rows, in place
      __syncthreads();
      for (int e = tid; e < npos * 8; e += NT) {
        const unsigned pos = (unsigned)e >> 3, slot = (unsigned)e & 7u, pc = slot ^ (pos & 7u);
        const bool real = pos >= (unsigned)U || (int)pos < nown;
        u32x4 v = *reinterpret_cast<u32x4*>(smem + pos * 128 + slot * 16);
        if (real) {
          u32x4 o;
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int c0 = (int)pc * 8 + 2 * qq;
            const f32x2 z = __builtin_elementwise_fma(f32x2{__uint_as_float(v[qq] << 16), __uint_as_float(v[qq] & 0xFFFF0000u)},
                                                      f32x2{p.in_scale[c0], p.in_scale[c0 + 1]}, f32x2{p.in_shift[c0], p.in_shift[c0 + 1]});
            o[qq] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack2(z[0], z[1])), s16x2{0, 0}));
          }
          *reinterpret_cast<u32x4*>(smem + pos * 128 + slot * 16) = o;
        }
      }
    }
    // the wave's rulebook rows: lane (fi, fh) contracts rows 64 wv + 32 t + fi
    uint32_t rb[MT][9];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int64_t r = row0 + wv * (32 * MT) + t * 32 + fi;
#pragma unroll
      for (int q = 0; q < 9; ++q) rb[t][q] = r < p.n ? p.lrb[r * 9 + q] : 0x3FFFFFFFu;      // past the end: every tap reads the zero row
    }
    __syncthreads();

    f32x16 acc[MT][2];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][nb][i] = 0.f;

    // B fragments (the tap's 8 KB of weights, read by every wave straight from L1 / L2) are requested ONE TAP ahead -- an L2 hit takes
    // longer than the 8 MFMAs of a half-tap step; A fragments come from the stage one step (half a tap) ahead.
    u32x4 A[2][MT][2], B[3][2][4];
    auto issue_b = [&](int k, int b_) __attribute__((always_inline)) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int s = 0; s < 4; ++s)       // fragment order of tl_pack_weight_frag: vector ((((k * CB + cb) * CH + ch) * 2 + j) * 64 + lane), CB = CH = 2
          B[b_][nb][s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, ((((k * 2 + nb) * 2 + (s >> 1)) * 2 + (s & 1)) * 64) * 16, 0));
    };
    auto issue_a = [&](int st, int a_) __attribute__((always_inline)) {
      const int k = st >> 1, h = st & 1;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const unsigned pos = (rb[t][k / 3] >> (10 * (k % 3))) & 1023u;
#pragma unroll
        for (int j = 0; j < 2; ++j) A[a_][t][j] = lds_r128(lds0 + st_off(pos, (unsigned)(2 * (2 * h + j) + fh)));
      }
    };
    issue_b(0, 0);
    if (TAPS > 1) issue_b(1, 1);
    issue_a(0, 0);
#pragma unroll
    for (int k = 0; k < TAPS; ++k) {
      const int b_ = k % 3;
      __builtin_amdgcn_sched_barrier(0);
      // the tap's weights were requested TWO taps ago (an L2 hit outlasts one tap of a single wave); the next tap's 8 loads may stay in flight
      if (k + 1 < TAPS) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(B[b_][nb][s]));
      if (k + 2 < TAPS) issue_b(k + 2, (k + 2) % 3);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int st = 2 * k + h, a_ = st & 1;
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < 2 * TAPS) { issue_a(st + 1, a_ ^ 1); asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * MT) : "memory"); } else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(A[a_][t][j]));            // (the MFMAs below must not move above the wait)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
              acc[t][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[a_][t][j]), __builtin_bit_cast(bf16x8, B[b_][nb][2 * h + j]), acc[t][nb], 0, 0, 0);
      }
    }
    __syncthreads();                                                     // every wave is done reading the stage: it becomes the epilogue buffer
    // ---- epilogue: the wave's (32 MT) x 64 fp32 tile through its own 26 KB of the stage, rows out as 128-B lines
    constexpr int WR = 32 * MT;
    float* ew = reinterpret_cast<float*>(smem) + wv * WR * 68;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) ew[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 68 + nb * 32 + fi] = acc[t][nb][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < WR / 8; ++i) {
      const int rr = i * 8 + (lane >> 3), pc = lane & 7;
      const int64_t r = row0 + wv * WR + rr;
      if (r < p.n && wv * WR + rr < nown && !(p.mode & 4)) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(ew + rr * 68 + pc * 8), v1 = *reinterpret_cast<const f32x4*>(ew + rr * 68 + pc * 8 + 4);
        const u32x4 o = {pack2(v0[0], v0[1]), pack2(v0[2], v0[3]), pack2(v1[0], v1[1]), pack2(v1[2], v1[3])};
        *reinterpret_cast<u32x4*>(p.out + r * 64 + pc * 8) = o;
      }
    }
    if (tid < 8) *reinterpret_cast<u32x4*>(smem + ZERO * 128 + tid * 16) = u32x4{0u, 0u, 0u, 0u};   // (the epilogue of wave 5 ends below the zero row; kept for safety)
  }
}

}  // namespace

extern "C" int proto_l2_conv(const void* x, const void* wfrag, void* out, const int32_t* halo, const int32_t* nhalo, const uint32_t* lrb,
                             const float* in_scale, const float* in_shift, int64_t n, int units, void* stream, int mode) {
  P p{(const uint16_t*)x, (const uint16_t*)wfrag, (uint16_t*)out, halo, nhalo, lrb, in_scale, in_shift, n, units, mode};
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_l2<27>), hipFuncAttributeMaxDynamicSharedMemorySize, POS * 128) != hipSuccess) return -2;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_l2<1>), hipFuncAttributeMaxDynamicSharedMemorySize, POS * 128) != hipSuccess) return -2;
    attr = true;
  }
  const int grid = units < 256 ? units : 256;
  if (mode & 2) k_conv_l2<1><<<grid, NT, POS * 128, (hipStream_t)stream>>>(p);
  else k_conv_l2<27><<<grid, NT, POS * 128, (hipStream_t)stream>>>(p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
