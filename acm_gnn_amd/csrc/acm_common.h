// Internal definitions shared by the translation units of libacm_hip.so.
// gfx950 (MI355X / CDNA4) only: wave = 64 lanes, 8 XCDs, block b lands on XCD b % 8.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "acm_hip.h"

#define ACM_WAVE 64
#define ACM_NXCD 8
#define ACM_WINDOW 16         // work items per window of the piece region (see build_items)
#define ACM_MIN_CHUNK 128
#define ACM_MAX_CHUNK 1024
#define ACM_GROUPS_IN_FLIGHT 8192  // 256 CUs x 8 waves x four 16-lane groups
#define ACM_LN_EPS 1e-5f

void acm_set_error(const char* fmt, ...);
// The process-wide tuning record (acm_hip.h: acm_tuning_t; defined in acm_csr.cpp).  Dispatch code reads it through this
// accessor -- never the environment.
const acm_tuning_t& acm_tuning();
#define ACM_ROWS16_EPI 1
#define ACM_ROWS16_BWD 2
#define ACM_ROWS16_LOCAL 4
#define ACM_GEMM_ROWS 1
#define ACM_GEMM_BX3 2
#define ACM_GEMM_BX3_WIDE 4
#define ACM_GEMM_ROWS_ALWAYS 8

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
// row and its owner runs the epilogue, otherwise it is one piece of a long row and
// its partial sums go to partial slot `slot` (combined in order by the fix-up pass, or
// through LDS by a kernel that takes whole windows, see build_items in acm_csr.cpp).
struct AcmItem {
    int32_t row, begin, end, slot;
};
// A long row and its partial slots [slot_begin, slot_end).
struct AcmLongRow {
    int32_t row, slot_begin, slot_end, windows;   // windows > 1: the row's pieces fill that many whole windows (acm_csr.cpp)
};

// Per-wave id streams of a pattern-only operator (acm_csr_build_streams, acm_csr.cpp): the column ids laid out in the
// order a wave of the streamed aggregate-first kernel consumes them.  A SLICE = four work items (rows, or pieces of at
// most `lmax` neighbours of a long row) of similar length, one per 16-lane group of a wave; it takes
// steps = ceil(longest / 32) wave steps of 32 neighbours per group; a step is 128 ids, [group][lane pair][4], padded with
// ACM_STREAM_SENTINEL (an id whose row offset lies beyond any table: a buffer load answers it with zeros, no memory access).
// The slices are dealt to `n_waves` waves (longest first, each to the least loaded wave); a wave's slices are contiguous
// in every array, so it walks ONE linear id stream and its descriptors come through the scalar unit.
#define ACM_STREAM_SENTINEL 0x07FFFFFF
#define ACM_STREAM_PAD_STEPS 8
struct AcmStreams {
    int32_t* ids;           // device, (total_steps + ACM_STREAM_PAD_STEPS) * 128
    int32_t* waves;         // device, n_waves x {slice_begin, slice_end, first_step, total_steps}
    int32_t* items;         // device, (n_slices + 2) x 4 groups x {row, slot, steps of the slice, 0}; row < 0: no item,
                            // slot < 0: a whole row
    AcmLongRow* long_rows;  // device: rows cut into pieces, with their partial slots
    int32_t* long_index;    // device, n_rows: index into long_rows or -1 (NULL without long rows)
    int32_t* counters;      // device, n_long arrival counters (zero between launches: the last arriver resets its own)
    float* slots;           // device, n_slots x 8 partial sums
    int64_t total_steps, n_slices, n_long, n_slots;
    int32_t n_waves, lmax;
};

// Per-wave BATCH streams over the handle's own work items (acm_csr_build_item_streams, acm_csr.cpp) for kernels that take one
// item at a time, 32 neighbours per wave step (acm_conv_acmii_v.hip).  The item list is cut into QUADS of four consecutive
// items (a wave finishes four rows together); quads are dealt longest-first to the least loaded of `n_waves` persistent
// waves; a wave's quads and their column ids are contiguous: batch b of the stream = ids[32 b .. 32 b + 31], idle slots and
// ACM_ITEM_STREAM_PAD batches behind the last wave hold n_cols (the index of a zero row the caller appends to its table).
#define ACM_ITEM_STREAM_PAD 8
struct AcmItemStreams {
    int32_t* ids;           // device, (total_batches + ACM_ITEM_STREAM_PAD) * 32
    int32_t* quads;         // device, n_quads x 4 items x {row, slot, batches, flags}; flags: 1 = item exists, 2 = it runs the row's
                            // own terms (a whole row, or the first piece of a long one)
    int32_t* waves;         // device, n_waves x {quad_begin, quad_end, first_batch, batches}
    int64_t total_batches, n_quads;
    int32_t n_waves;
};

struct acm_csr {
    int64_t n_rows, n_cols, nnz;
    int32_t chunk, max_degree;
    int32_t* indptr;   // device, n_rows + 1
    int32_t* indices;  // device, nnz
    float* vals;       // device, nnz
    int32_t* src_pos;  // device, nnz (transposed handles only): index into the source's vals
    AcmItem* items;    // device
    int64_t n_items;
    AcmLongRow* long_rows;  // device
    int32_t* long_index;    // device, n_rows: index into long_rows, or -1 (NULL when there are no long rows)
    int64_t n_long;
    int64_t n_slots;
    int64_t n_windows;      // the first n_windows * ACM_WINDOW items are the pieces of the long rows
    int64_t n_multi;        // long rows that take several windows (their window sums are added by a second launch)
    AcmStreams* streams;    // NULL until acm_csr_build_streams
    AcmItemStreams* item_streams;   // NULL until acm_csr_build_item_streams
    int device;
};

struct StreamView {
    const int32_t* ids;
    const int32_t* waves;
    const int32_t* items;
    const AcmLongRow* long_rows;
    const int32_t* long_index;
    int32_t* counters;
    float* slots;
    unsigned ids_bytes, slots_bytes;
    int n_waves;
};

// Device-side view handed to kernels by value.
struct CsrView {
    const AcmItem* items;
    int n_items;
    const AcmLongRow* long_rows;
    const int32_t* long_index;
    int n_long;
    int n_windows;
    int n_multi;
    const int32_t* indices;
    const float* vals;
};

// Second phase of a partial-sum reduction: launch `segs` now (defer == NULL) or append them to the caller's list.
int acm_reduce_emit(acm_reduce_list_t* defer, const acm_reduce_seg_t* segs, int n, hipStream_t st);
int acm_reduce_check_segment(const acm_reduce_seg_t& sg, int index);      // ACM_OK or ACM_EINVAL (+ message)

static inline CsrView acm_view(const acm_csr* a) {
    CsrView v;
    v.items = a->items;
    v.n_items = (int)a->n_items;
    v.long_rows = a->long_rows;
    v.long_index = a->long_index;
    v.n_long = (int)a->n_long;
    v.n_windows = (int)a->n_windows;
    v.n_multi = (int)a->n_multi;
    v.indices = a->indices;
    v.vals = a->vals;
    return v;
}

#ifdef __HIPCC__
// ---- device helpers ---------------------------------------------------------
// All-reduce (sum) over the W consecutive lanes that share lane / W; W power of two <= 64.
// Pure VALU: DPP quad_perm (xor 1, xor 2), row_half_mirror / row_mirror (the partner of lane i
// is 7-i / 15-i, which after the earlier steps holds the other half's total), then gfx950's
// v_permlane16_swap / v_permlane32_swap for the row-crossing steps.  ~2 instructions per step,
// no LDS crossbar (ds_bpermute) and no s_waitcnt, fixed summation tree => deterministic.
// Every lane of the group must be active.
template <int CTRL>
__device__ __forceinline__ float acm_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
typedef unsigned acm_u32x2 __attribute__((ext_vector_type(2)));
template <int W>
__device__ __forceinline__ float acm_group_sum(float v) {
    static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16 || W == 32 || W == 64, "group width");
    if (W >= 2) v += acm_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
    if (W >= 4) v += acm_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
    if (W >= 8) v += acm_dpp<0x141>(v);   // row_half_mirror
    if (W >= 16) v += acm_dpp<0x140>(v);  // row_mirror
    if (W >= 32) {
        const acm_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    if (W >= 64) {
        const acm_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v;
}

// Sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48); result in every lane.
__device__ __forceinline__ float acm_cross_row_sum(float v) {
    acm_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// lane u of each 16-lane row -> every lane of that row (v_mov_b32_dpp row_newbcast:u); u is a constant after unrolling,
// so the switch folds
__device__ __forceinline__ int acm_row_bcast(int v, int u) {
#define ACM_BC(U) case U: return __builtin_amdgcn_update_dpp(0, v, 0x150 + U, 0xf, 0xf, false)
    switch (u & 15) {
        ACM_BC(0); ACM_BC(1); ACM_BC(2); ACM_BC(3); ACM_BC(4); ACM_BC(5); ACM_BC(6); ACM_BC(7);
        ACM_BC(8); ACM_BC(9); ACM_BC(10); ACM_BC(11); ACM_BC(12); ACM_BC(13); ACM_BC(14);
        default: return __builtin_amdgcn_update_dpp(0, v, 0x15F, 0xf, 0xf, false);
    }
#undef ACM_BC
}

__device__ __forceinline__ int acm_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float acm_lane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// One row of the masked NLL (acm_nll_loss): di = wi (softmax(zi) - onehot(yi)); returns wi (logsumexp(zi) - zi[yi]).
// Rows outside the training set (wi == 0) get a zero gradient and no loss.
__device__ __forceinline__ float acm_nll_row(int C, const float* zi, int yi, float wi, float* di) {
    if (wi == 0.f) {
        for (int c = 0; c < C; ++c) di[c] = 0.f;
        return 0.f;
    }
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, zi[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(zi[c] - m);
    const float lse = m + logf(s);
    const float inv = 1.0f / s;
    const float zy = zi[yi];
    for (int c = 0; c < C; ++c) di[c] = wi * (expf(zi[c] - m) * inv - (c == yi ? 1.f : 0.f));
    return wi * (lse - zy);
}

// ------------------------------------------------------------------ counter-based dropout (acm_dropout_t)
struct AcmDropCtx {
    unsigned k0, k1, c2, c3, thresh, tag16;
    float inv_keep;
    long row_offset;
    bool on;
};

__device__ __forceinline__ AcmDropCtx acm_drop_ctx(const acm_dropout_t& d) {
    AcmDropCtx c;
    c.on = d.p > 0.f;
    const unsigned long long step = c.on ? (unsigned long long)(d.step[0] + d.step_offset) : 0ull;
    c.k0 = (unsigned)d.seed;
    c.k1 = (unsigned)(d.seed >> 32);
    c.c2 = (unsigned)step;
    c.c3 = (unsigned)(step >> 32);
    const double t = (double)d.p * 4294967296.0;
    c.thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
    c.tag16 = ((unsigned)d.tag) << 16;
    c.inv_keep = 1.0f / (1.0f - d.p);
    c.row_offset = (long)d.row_offset;
    return c;
}

// Philox4x32 with 7 rounds (Crush-resistant per Salmon et al., SC'11): four 32-bit words per (row, block)
__device__ __forceinline__ void acm_philox7(const AcmDropCtx& c, long row, int block, unsigned (&w)[4]) {
    unsigned x0 = (unsigned)(row + c.row_offset), x1 = (unsigned)block | c.tag16, x2 = c.c2, x3 = c.c3;
    unsigned k0 = c.k0, k1 = c.k1;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        // one 32 x 32 -> 64-bit product per multiplier (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair:
        // 32-bit integer multiplies are quarter rate, and 28 of them per call were 11 % of agg_bwd_kernel
        const unsigned long long q0 = (unsigned long long)0xD2511F53u * x0, q1 = (unsigned long long)0xCD9E8D57u * x2;
        const unsigned hi0 = (unsigned)(q0 >> 32), lo0 = (unsigned)q0;
        const unsigned hi1 = (unsigned)(q1 >> 32), lo1 = (unsigned)q1;
        x0 = hi1 ^ x1 ^ k0;
        x1 = lo1;
        x2 = hi0 ^ x3 ^ k1;
        x3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    w[0] = x0, w[1] = x1, w[2] = x2, w[3] = x3;
}

// factor (0 or 1/(1-p)) of one element
__device__ __forceinline__ float acm_drop1(const AcmDropCtx& c, long row, int col) {
    if (!c.on) return 1.f;
    unsigned w[4];
    acm_philox7(c, row, (col & 15) + 16 * (col >> 6), w);
    const int q = (col >> 4) & 3;
    const unsigned v = q == 0 ? w[0] : (q == 1 ? w[1] : (q == 2 ? w[2] : w[3]));
    return v >= c.thresh ? c.inv_keep : 0.f;
}

// factors of columns m, m + 16, m + 32, m + 48 (the grouped layout's lane; F <= 64): one Philox call
__device__ __forceinline__ void acm_drop4(const AcmDropCtx& c, long row, int m, float (&f)[4]) {
    if (!c.on) {
        f[0] = f[1] = f[2] = f[3] = 1.f;
        return;
    }
    unsigned w[4];
    acm_philox7(c, row, m, w);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = w[i] >= c.thresh ? c.inv_keep : 0.f;
}

#endif
