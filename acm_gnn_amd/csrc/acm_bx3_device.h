// Split-bf16 operands of v_mfma_f32_16x16x32_bf16 (acm_gemm_bx3.hip explains the arithmetic: an fp32 number is exactly the
// sum of three bf16 numbers, six of the nine partial products are kept), shared by the projections (acm_gemm_bx3.hip) and the
// wide aggregate-first layer (acm_conv_aggw.hip).
#pragma once
#include "acm_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bitsf(unsigned x) { return __builtin_bit_cast(float, x); }
// two fp32 -> their upper halves as one dword (element 2t in the low half)
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// eight fp32 -> three vectors of eight bf16 (hi, mid, lo), x = hi + mid + lo exactly
__device__ __forceinline__ void split3(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float a = x[2 * t], b = x[2 * t + 1];
        const float ra = a - bitsf(fbits(a) & 0xFFFF0000u), rb = b - bitsf(fbits(b) & 0xFFFF0000u);
        const float sa = ra - bitsf(fbits(ra) & 0xFFFF0000u), sb = rb - bitsf(fbits(rb) & 0xFFFF0000u);
        hi[t] = pack_hi16(fbits(a), fbits(b));
        mid[t] = pack_hi16(fbits(ra), fbits(rb));
        lo[t] = pack_hi16(fbits(sa), fbits(sb));
    }
}

__device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// column of X / row of W behind contraction slot (k block kb, lane group g, element e)
__device__ __forceinline__ int bx3_k(int kb, int g, int e) { return 64 * (kb >> 1) + 16 * (2 * (kb & 1) + (e >> 2)) + 4 * g + (e & 3); }

}  // namespace
