// Arithmetic modes of the MFMA kernels (convolutions, Linears, the fused level tail).  Every such kernel is a template on
//   NP = 3   an fp32 multiply-add as THREE bf16 products, lo*hi + hi*lo + hi*hi with fp32 accumulation (operands split
//            v = hi + lo in bfloat16): fp32-faithful to ~5e-6 of the tensor scale -- the default and the headline;
//   NP = 1   ONE bf16 product (both operands rounded to bfloat16): opt-in, ~2^-8 relative per operand;
//   NP = 2   ONE fp16 product on v_mfma_f32_32x32x16_f16 (same rate as the bf16 instruction, 11 instead of 8 significant bits):
//            opt-in, the twin of the reference's fp16 operator (TU/multi_scale_3ddeformable_attn_function.py:353-428; BASELINE.json
//            config #5 "fp16").  Operands are rounded to IEEE half and SATURATED at +-65504 (an fp32 activation beyond the half
//            range becomes the largest half, not infinity); a NaN stays a NaN.
// The 16-bit operands travel in `__bf16`-typed containers whatever the mode (planes, fragments, LDS images are byte-identical in
// layout); only the rounding on the way in (op_hi) and the instruction that multiplies them (mma_hh) know the format.
// sgc_set_conv_products(3 | 1 | 2) selects the mode process-wide (include/sgcdet_amd.h).
#pragma once
#include <hip/hip_runtime.h>

namespace sgc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// the operand (NP = 1, 2) or its high part (NP = 3)
template <int NP>
__device__ __forceinline__ __bf16 op_hi(float v) {
  if constexpr (NP == 2) {
    const float c = v != v ? v : fminf(fmaxf(v, -65504.f), 65504.f);
    return __builtin_bit_cast(__bf16, (_Float16)c);
  } else {
    return (__bf16)v;
  }
}
// the low part of the NP = 3 split; unused (zero bits) in the one-product modes
template <int NP>
__device__ __forceinline__ __bf16 op_lo(float v, __bf16 hi) {
  if constexpr (NP == 3) return (__bf16)(v - (float)hi);
  else return __builtin_bit_cast(__bf16, (unsigned short)0);
}
// the hi * hi product (the only one of the one-product modes)
template <int NP>
__device__ __forceinline__ f32x16 mma_hh(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (NP == 2)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

}  // namespace sgc
