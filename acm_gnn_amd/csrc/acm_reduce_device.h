// The second phase of a partial-sum reduction as a device function (shared by reduce_segments_kernel, acm_reduce.hip, and
// the optimizer kernel that flushes the step's pending reductions itself, acm_optim.hip): block `e` of segment `sg`
// sums its output element(s) -- thread t of 256 adds blocks t, t + 256, ..., then a binary tree -- and hands every result to
// `emit(destination pointer, value)`.  A segment whose slabs are stored in groups of 32 elements -- partial[group][block][32]:
// row_stride 32, elem_stride = distance between groups, q0 a multiple of 32 -- is summed by one block per GROUP: thread t
// adds the 128-byte lines of blocks t, t + 256, ... (one line per block instead of one float out of each of 32 lines).
// Per element the order of additions is the same in both forms: bit-identical results.
#pragma once
#include "acm_common.h"

#define ACM_REDUCE_LDS (32 * 256)      /* floats of shared memory the caller provides */

__device__ __forceinline__ bool acm_seg_by_lines(const acm_reduce_seg_t& sg) {
    return sg.elem_stride > 0 && sg.row_stride == 32 && sg.q0 % 32 == 0;
}
// blocks a segment takes
static inline int acm_seg_blocks(const acm_reduce_seg_t& sg) {
    const bool lines = sg.elem_stride > 0 && sg.row_stride == 32 && sg.q0 % 32 == 0;
    return lines ? (sg.len + 31) / 32 : sg.len;
}

template <class Emit>
__device__ __forceinline__ void acm_reduce_block(const acm_reduce_seg_t& sg, int e, float* red, Emit&& emit) {
    if (acm_seg_by_lines(sg)) {
        const float* __restrict__ base = sg.partial + (long)(sg.q0 / 32 + e) * sg.elem_stride;
        float s[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) s[k] = 0.f;
        for (int b = threadIdx.x; b < sg.nblk; b += 256) {
            const float4* ln = reinterpret_cast<const float4*>(base + (long)b * 32);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 v = ln[k];
                s[4 * k] += v.x, s[4 * k + 1] += v.y, s[4 * k + 2] += v.z, s[4 * k + 3] += v.w;
            }
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) red[k * 256 + threadIdx.x] = s[k];
        __syncthreads();
        for (int m = 128; m >= 1; m >>= 1) {
            if ((int)threadIdx.x < m) {
#pragma unroll
                for (int k = 0; k < 32; ++k) red[k * 256 + threadIdx.x] += red[k * 256 + threadIdx.x + m];
            }
            __syncthreads();
        }
        const int el = e * 32 + (int)threadIdx.x;
        if (threadIdx.x < 32 && el < sg.len) {
            const int j = el / sg.inner, q = el % sg.inner;
            const long col = sg.col_block ? (long)(q / sg.col_block) * sg.block_stride + q % sg.col_block : q;
            emit(sg.dst + (long)j * sg.outer_stride + col, red[threadIdx.x * 256]);
        }
        return;
    }
    const int qq = sg.q0 + e;
    const float* __restrict__ src = sg.elem_stride > 0 ? sg.partial + (long)(qq / sg.row_stride) * sg.elem_stride + qq % sg.row_stride
                                                       : sg.partial + qq;
    float s = 0.f;
    for (int b = threadIdx.x; b < sg.nblk; b += 256) s += src[(long)b * sg.row_stride];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int j = e / sg.inner, q = e % sg.inner;
        const long col = sg.col_block ? (long)(q / sg.col_block) * sg.block_stride + q % sg.col_block : q;
        emit(sg.dst + (long)j * sg.outer_stride + col, red[0]);
    }
}
