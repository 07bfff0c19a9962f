// bf16 plane helpers of the fast-precision kernels (niw_mlp_fast.hip, the fast path of niw_dw_gemm.hip).
#pragma once
#include "niw_common.h"

namespace niw {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// two fp32 values -> one dword of two bf16 (round to nearest even, NaN kept: v_cvt_pk_bf16_f32); element 0 in the low half
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
// (hi, mid) planes of a pair: hi = bf16(x), mid = bf16(x - hi); x - hi is exact in fp32 (hi keeps the leading 8 significand bits)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid) {
    hi = pack_bf16(a, b);
    const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = pack_bf16(a - ha, b - hb);
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4_t& a, const u32x4_t& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

}  // namespace niw
