// Second phase of every partial-sum reduction of libacm_hip.so (acm_reduce_seg_t, include/acm_hip.h): one block per
// output element, thread t adds blocks t, t + 256, ... of its column, binary tree over the 256 threads, one store.
// Up to 32 segments (different workspaces, different destinations) share a launch; a call without a deferral list
// launches its own segments at once through the same kernel.
#include "acm_common.h"

namespace {

constexpr int SEGS_PER_LAUNCH = 32;

struct ReducePack {
    int n;
    int first[SEGS_PER_LAUNCH + 1];      // first[i] = index of segment i's first block
    acm_reduce_seg_t seg[SEGS_PER_LAUNCH];
};

// A segment whose slabs are stored in groups of 32 elements -- partial[group][block][32]: row_stride 32, elem_stride =
// distance between groups, q0 a multiple of 32 -- is summed by one block per GROUP: thread t adds the 128-byte lines of
// blocks t, t + 256, ... (one line per block instead of one float out of each of 32 lines), then the same binary tree over
// the 256 threads for each of the 32 elements.  Per element the order of additions is exactly that of the one-block-per-
// element form, so the two are bit-identical.
__device__ __forceinline__ bool seg_by_lines(const acm_reduce_seg_t& sg) {
    return sg.elem_stride > 0 && sg.row_stride == 32 && sg.q0 % 32 == 0;
}

__global__ __launch_bounds__(256) void reduce_segments_kernel(ReducePack pk) {
    __shared__ __attribute__((aligned(16))) float red[32 * 256];
    int e = blockIdx.x, i = 0;
    while (i + 1 < pk.n && e >= pk.first[i + 1]) ++i;        // block-uniform
    const acm_reduce_seg_t& sg = pk.seg[i];
    e -= pk.first[i];
    if (seg_by_lines(sg)) {
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
            sg.dst[(long)j * sg.outer_stride + col] = red[threadIdx.x * 256];
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
        sg.dst[(long)j * sg.outer_stride + col] = red[0];
    }
}

int launch_segments(const acm_reduce_seg_t* segs, int n, hipStream_t st) {
    for (int base = 0; base < n; base += SEGS_PER_LAUNCH) {
        ReducePack pk;
        pk.n = 0;
        int blocks = 0;
        for (int i = base; i < n && pk.n < SEGS_PER_LAUNCH; ++i) {
            const acm_reduce_seg_t& sg = segs[i];
            ACM_REQUIRE(sg.partial && sg.dst && sg.nblk >= 1 && sg.len >= 0 && sg.inner >= 1 && sg.q0 >= 0 &&
                            (sg.elem_stride > 0 ? (sg.row_stride >= 1 && (int64_t)sg.elem_stride >= (int64_t)sg.nblk * sg.row_stride)
                                                : sg.row_stride >= sg.q0 + sg.len), ACM_EINVAL,
                        "acm_reduce: malformed segment %d (nblk %d, row_stride %d, q0 %d, len %d, inner %d)", i,
                        sg.nblk, sg.row_stride, sg.q0, sg.len, sg.inner);
            if (sg.len == 0) continue;
            pk.first[pk.n] = blocks;
            pk.seg[pk.n] = sg;
            const bool lines = sg.elem_stride > 0 && sg.row_stride == 32 && sg.q0 % 32 == 0;
            ACM_REQUIRE(!lines || (((uintptr_t)sg.partial) % 16 == 0 && sg.elem_stride % 4 == 0), ACM_EINVAL,
                        "acm_reduce: grouped segment %d must be 16-byte aligned", i);
            blocks += lines ? (sg.len + 31) / 32 : sg.len;
            ++pk.n;
        }
        pk.first[pk.n] = blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(reduce_segments_kernel, dim3(blocks), dim3(256), 0, st, pk);
        ACM_CHECK_HIP(hipGetLastError());
    }
    return ACM_OK;
}

}  // namespace

int acm_reduce_emit(acm_reduce_list_t* defer, const acm_reduce_seg_t* segs, int n, hipStream_t st) {
    if (!defer) return launch_segments(segs, n, st);
    ACM_REQUIRE(defer->segs && defer->n >= 0 && defer->n + n <= defer->cap, ACM_ENOMEM,
                "acm_reduce: deferral list full (%d + %d segments, capacity %d)", defer->n, n, defer->cap);
    for (int i = 0; i < n; ++i) defer->segs[defer->n++] = segs[i];
    return ACM_OK;
}

extern "C" int acm_reduce_flush(acm_reduce_list_t* list, acm_stream_t stream) {
    ACM_REQUIRE(list && (list->n == 0 || list->segs) && list->n >= 0 && list->n <= list->cap, ACM_EINVAL,
                "acm_reduce_flush: NULL or inconsistent list");
    const int st = launch_segments(list->segs, list->n, (hipStream_t)stream);
    if (st == ACM_OK) list->n = 0;
    return st;
}
