// Helpers of the sixteen-rows-per-wave row-local kernels (acm_conv_agg16.hip, acm_conv_local16.hip): lane (g, m) of a wave
// holds columns 16 t + 4 g + r (t, r = 0..3) of row m, so a whole row sits in the four lanes m, m + 16, m + 32, m + 48.
#pragma once
#include "acm_conv_device.h"

namespace {

// sum over the four lanes that hold one row (lanes m, m + 16, m + 32, m + 48); result in all four
__device__ __forceinline__ float row4_sum(float v) { return acm_cross_row_sum(v); }

// sum over the 16 lanes of a row for 16 values per lane, leaving value i's total in lane i (m = i): a reduce-scatter of four
// DPP exchange steps (partner = 15 - m, 7 - m within the half, m ^ 2, m ^ 1; each lane keeps the half of the values its
// own lane bit selects and adds the partner's copy of them) -- 45 instructions, where sixteen all-reduces cost 64 and
// sixteen per-lane accumulators would pin 48 registers per kernel.
__device__ __forceinline__ float row_reduce_scatter16(const float (&v)[16], int m) {
    const bool b3 = (m & 8) != 0, b2 = (m & 4) != 0, b1 = (m & 2) != 0, b0 = (m & 1) != 0;
    float a[8], b[4], c[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (b3 ? v[i + 8] : v[i]) + acm_dpp<0x140>(b3 ? v[i] : v[i + 8]);       // row_mirror
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = (b2 ? a[i + 4] : a[i]) + acm_dpp<0x141>(b2 ? a[i] : a[i + 4]);       // row_half_mirror
#pragma unroll
    for (int i = 0; i < 2; ++i) c[i] = (b1 ? b[i + 2] : b[i]) + acm_dpp<0x4E>(b1 ? b[i] : b[i + 2]);        // quad_perm [2,3,0,1]
    return (b0 ? c[1] : c[0]) + acm_dpp<0xB1>(b0 ? c[0] : c[1]);                                            // quad_perm [1,0,3,2]
}


}  // namespace
