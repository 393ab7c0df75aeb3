// Internal definitions shared by the translation units of libacm_hip.so.
// gfx950 (MI355X / CDNA4) only: wave = 64 lanes, 8 XCDs, block b lands on XCD b % 8.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acm_hip.h"

#define ACM_WAVE 64
#define ACM_NXCD 8
#define ACM_DEFAULT_CHUNK 256
#define ACM_LN_EPS 1e-5f

void acm_set_error(const char* fmt, ...);

#define ACM_CHECK_HIP(expr)                                                            \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) {                                                       \
            acm_set_error("%s -> %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                          __LINE__);                                                   \
            return ACM_EHIP;                                                           \
        }                                                                              \
    } while (0)

#define ACM_REQUIRE(cond, code, ...)    \
    do {                                \
        if (!(cond)) {                  \
            acm_set_error(__VA_ARGS__); \
            return (code);              \
        }                               \
    } while (0)

// A work item: neighbours [begin,end) of `row`; slot < 0 => the item is the whole
// row and its owner runs the epilogue, otherwise it is one chunk of a long row and
// its partial sums go to partial slot `slot` (combined in order by the fix-up pass).
struct AcmItem {
    int32_t row, begin, end, slot;
};
// A long row and its partial slots [slot_begin, slot_end).
struct AcmLongRow {
    int32_t row, slot_begin, slot_end, pad;
};

struct acm_csr {
    int64_t n_rows, n_cols, nnz;
    int32_t chunk, max_degree;
    int32_t* indptr;   // device, n_rows + 1
    int32_t* indices;  // device, nnz
    float* vals;       // device, nnz
    AcmItem* items;    // device
    int64_t n_items;
    AcmLongRow* long_rows;  // device
    int64_t n_long;
    int64_t n_slots;
    int device;
};

// Device-side view handed to kernels by value.
struct CsrView {
    const AcmItem* items;
    int n_items;
    const AcmLongRow* long_rows;
    int n_long;
    const int32_t* indices;
    const float* vals;
};

static inline CsrView acm_view(const acm_csr* a) {
    CsrView v;
    v.items = a->items;
    v.n_items = (int)a->n_items;
    v.long_rows = a->long_rows;
    v.n_long = (int)a->n_long;
    v.indices = a->indices;
    v.vals = a->vals;
    return v;
}

#ifdef __HIPCC__
// ---- device helpers ---------------------------------------------------------
// Remap the linear block id so that each XCD (private 4 MiB L2) walks a contiguous
// range of work items: block b runs on XCD b % 8 (observed dispatch order; speed
// only, never correctness).  Bijective for any grid size.
__device__ __forceinline__ int acm_xcd_swizzle(int b, int nblk) {
    const int xcd = b % ACM_NXCD, idx = b / ACM_NXCD;
    const int q = nblk / ACM_NXCD, r = nblk % ACM_NXCD;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// All-reduce (sum) over the W consecutive lanes that share lane / W; W power of two <= 64.
template <int W>
__device__ __forceinline__ float acm_group_sum(float v) {
#pragma unroll
    for (int m = W / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ int acm_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float acm_lane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
#endif
