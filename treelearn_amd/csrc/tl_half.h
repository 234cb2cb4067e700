// The 16-bit storage type of a translation unit: bfloat16 (default) or, when the unit is compiled with -DTL_F16_BUILD, IEEE half.
// The conv / head kernels treat their "16-bit" flag as a storage width and go through these four helpers for every conversion and
// for the matrix instruction, so the SAME kernel sources are compiled a second time for float16 (treelearn_amd/build.py builds
// tl_conv_direct / _stream / _streamq / _small / tl_head twice); in that build the launchers carry an _f16 suffix (tl_conv_internal.h)
// and tl_conv_fwd / tl_head_mlp route TL_F16 there.  BASELINE config 5 names fp16; the reference trains under fp16 autocast
// (tools/training/train.py:32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t h16_u32x4 __attribute__((ext_vector_type(4)));
typedef float h16_f32x16 __attribute__((ext_vector_type(16)));

#ifdef TL_F16_BUILD
typedef _Float16 h16_x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16_x8 __attribute__((ext_vector_type(8)));
static __device__ __forceinline__ uint32_t h16_pack2(float lo, float hi) {        // round-to-nearest-even (v_cvt_f16_f32); > 65504 -> inf
  const h16_x2 v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
static __device__ __forceinline__ float h16_lo(uint32_t u) { return (float)__builtin_bit_cast(h16_x2, u)[0]; }
static __device__ __forceinline__ float h16_hi(uint32_t u) { return (float)__builtin_bit_cast(h16_x2, u)[1]; }
static __device__ __forceinline__ h16_f32x16 h16_mfma(const h16_u32x4& a, const h16_u32x4& b, const h16_f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16_x8, a), __builtin_bit_cast(h16_x8, b), c, 0, 0, 0);
}
#else
typedef __bf16 h16_x8 __attribute__((ext_vector_type(8)));
typedef __bf16 h16_x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t h16_pack2(float lo, float hi) {        // round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950
  const h16_x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
static __device__ __forceinline__ float h16_lo(uint32_t u) { return __uint_as_float(u << 16); }
static __device__ __forceinline__ float h16_hi(uint32_t u) { return __uint_as_float(u & 0xFFFF0000u); }
static __device__ __forceinline__ h16_f32x16 h16_mfma(const h16_u32x4& a, const h16_u32x4& b, const h16_f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(h16_x8, a), __builtin_bit_cast(h16_x8, b), c, 0, 0, 0);
}
#endif
