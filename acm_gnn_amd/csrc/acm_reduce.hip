// Second phase of every partial-sum reduction of libacm_hip.so (acm_reduce_seg_t, include/acm_hip.h): one block per
// output element, thread t adds blocks t, t + 256, ... of its column, binary tree over the 256 threads, one store.
// Up to 32 segments (different workspaces, different destinations) share a launch; a call without a deferral list
// launches its own segments at once through the same kernel.
#include "acm_reduce_device.h"

namespace {

constexpr int SEGS_PER_LAUNCH = 32;

struct ReducePack {
    int n;
    int first[SEGS_PER_LAUNCH + 1];      // first[i] = index of segment i's first block
    acm_reduce_seg_t seg[SEGS_PER_LAUNCH];
};

__global__ __launch_bounds__(256) void reduce_segments_kernel(ReducePack pk) {
    __shared__ __attribute__((aligned(16))) float red[ACM_REDUCE_LDS];
    int e = blockIdx.x, i = 0;
    while (i + 1 < pk.n && e >= pk.first[i + 1]) ++i;        // block-uniform
    acm_reduce_block(pk.seg[i], e - pk.first[i], red, [](float* dst, float v) { *dst = v; });
}

int launch_segments(const acm_reduce_seg_t* segs, int n, hipStream_t st) {
    for (int base = 0; base < n; base += SEGS_PER_LAUNCH) {
        ReducePack pk;
        pk.n = 0;
        int blocks = 0;
        for (int i = base; i < n && pk.n < SEGS_PER_LAUNCH; ++i) {
            const acm_reduce_seg_t& sg = segs[i];
            const int ok = acm_reduce_check_segment(sg, i);
            if (ok != ACM_OK) return ok;
            if (sg.len == 0) continue;
            pk.first[pk.n] = blocks;
            pk.seg[pk.n] = sg;
            blocks += acm_seg_blocks(sg);
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

int acm_reduce_check_segment(const acm_reduce_seg_t& sg, int i) {
    ACM_REQUIRE(sg.partial && sg.dst && sg.nblk >= 1 && sg.len >= 0 && sg.inner >= 1 && sg.q0 >= 0 &&
                    (sg.elem_stride > 0 ? (sg.row_stride >= 1 && (int64_t)sg.elem_stride >= (int64_t)sg.nblk * sg.row_stride)
                                        : sg.row_stride >= sg.q0 + sg.len), ACM_EINVAL,
                "acm_reduce: malformed segment %d (nblk %d, row_stride %d, q0 %d, len %d, inner %d)", i,
                sg.nblk, sg.row_stride, sg.q0, sg.len, sg.inner);
    const bool lines = sg.elem_stride > 0 && sg.row_stride == 32 && sg.q0 % 32 == 0;
    ACM_REQUIRE(!lines || (((uintptr_t)sg.partial) % 16 == 0 && sg.elem_stride % 4 == 0), ACM_EINVAL,
                "acm_reduce: grouped segment %d must be 16-byte aligned", i);
    return ACM_OK;
}

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
