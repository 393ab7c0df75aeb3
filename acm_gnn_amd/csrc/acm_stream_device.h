// The narrow gather P = row_scale * (A xg) over per-wave id streams (acm_csr_build_streams), as a ROLE one or more waves of
// another kernel can take (acm_conv_agg.hip: agg_bwd_gather_kernel; acm_conv_agg16.hip: agg_bwd16_gather_kernel).
#pragma once
#include "acm_common.h"

typedef float acm_f32x4 __attribute__((ext_vector_type(4)));
typedef int acm_i32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- the gather alone, as a ROLE of another kernel
// One wave's share of  agg = row_scale * (A xg)  over the id streams (32-byte rows, pattern-only operator): the loop of
// agg_stream_kernel without the row-local stage.  acm_conv_agg_bwd runs it in two extra waves per workgroup for the NEXT
// step's first layer (acm_conv_agg_bwd_t.next_agg): the backward's waves keep the vector unit busy, these keep the memory
// system busy, and a kernel of each kind on two streams would not share the CUs.
struct GatherRole {
    StreamView sv;
    const float* xg;
    unsigned xg_bytes;
    const float* row_scale;
    float* agg;
    long ld_agg;
};

static __device__ __forceinline__ void stream_gather_role(const GatherRole& gr, const int W) {
    const StreamView& sv = gr.sv;
    const int lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15, e = gl >> 1, h = gl & 1;
    if (W >= sv.n_waves) return;
    int s = sv.waves[W * 4 + 0];
    const int s_end = sv.waves[W * 4 + 1];
    if (s >= s_end) return;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gr.xg), 0, gr.xg_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(sv.ids), 0, sv.ids_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(sv.slots, 0, sv.slots_bytes, 0x00020000);
    int ioff = sv.waves[W * 4 + 2] * 512 + (g * 8 + e) * 16;
    const int hoff = h * 16;
    acm_f32x4 za[4], zb[4];
#define ACM_ISSUE(Z, J)                                                                                         \
    do {                                                                                                        \
        Z[0] = __builtin_bit_cast(acm_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (J).x * 32 + hoff, 0, 0)); \
        Z[1] = __builtin_bit_cast(acm_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (J).y * 32 + hoff, 0, 0)); \
        Z[2] = __builtin_bit_cast(acm_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (J).z * 32 + hoff, 0, 0)); \
        Z[3] = __builtin_bit_cast(acm_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (J).w * 32 + hoff, 0, 0)); \
    } while (0)
#define ACM_IDS(OFF) __builtin_bit_cast(acm_i32x4, __builtin_amdgcn_raw_buffer_load_b128(ri, (OFF), 0, 0))
    // slice descriptors {row, slot, steps, -} per group, requested a slice ahead (as the row scale of the slice's rows)
    const acm_i32x4* items = reinterpret_cast<const acm_i32x4*>(sv.items);
    acm_i32x4 item = items[s * 4 + g];
    int rem = __builtin_amdgcn_readfirstlane(item.z);
    float rs_cur = gr.row_scale ? gr.row_scale[item.x >= 0 ? item.x : 0] : 1.f;
    acm_i32x4 item_next = items[(s + 1) * 4 + g];
    acm_i32x4 q0, q1;
    {
        // (scheduling barriers: the loop's counted waits assume exactly this issue order -- za, ids, zb, ids; if the
        // scheduler swaps the two independent row groups here, the loop head must wait for everything, every time)
        const acm_i32x4 j0 = ACM_IDS(ioff), j1 = ACM_IDS(ioff + 512);
        __builtin_amdgcn_sched_barrier(0);
        ACM_ISSUE(za, j0);
        q0 = ACM_IDS(ioff + 1024);
        __builtin_amdgcn_sched_barrier(0);
        ACM_ISSUE(zb, j1);
        q1 = ACM_IDS(ioff + 1536);
        __builtin_amdgcn_sched_barrier(0);
    }
    ioff += 2048;
    acm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto finish = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] += acm_dpp<0x4E>(acc[i]);     // quad_perm [2,3,0,1]
            acc[i] += acm_dpp<0x124>(acc[i]);    // row_ror:4
            acc[i] += acm_dpp<0x128>(acc[i]);    // row_ror:8
        }
        const int slot = item.y;
        bool active = item.x >= 0 && slot < 0;
        const int row = item.x >= 0 ? item.x : 0;
        if (__builtin_amdgcn_ballot_w64(slot >= 0) != 0ull) {      // some group holds a piece of a long row
            if (slot >= 0) {
                const int li = sv.long_index[row];
                const AcmLongRow lr = sv.long_rows[li];
                if (gl < 2)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(acm_i32x4, acc), rp, slot * 32 + hoff, 0, /*sc1*/ 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                int old = 0;
                if (gl == 0) old = __hip_atomic_fetch_add(sv.counters + li, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                old = acm_row_bcast(old, 0);
                if (old == lr.slot_end - lr.slot_begin - 1) {      // every other piece has arrived
                    if (gl == 0) __hip_atomic_store(sv.counters + li, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    acm_f32x4 tot = {0.f, 0.f, 0.f, 0.f};
                    for (int q = lr.slot_begin + e; q < lr.slot_end; q += 8)
                        tot += __builtin_bit_cast(acm_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, q * 32 + hoff, 0, /*sc1*/ 16));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        tot[i] += acm_dpp<0x4E>(tot[i]);
                        tot[i] += acm_dpp<0x124>(tot[i]);
                        tot[i] += acm_dpp<0x128>(tot[i]);
                    }
                    acc = tot;
                    active = true;
                }
            }
        }
        if (active && gl < 2) *reinterpret_cast<acm_f32x4*>(gr.agg + (long)row * gr.ld_agg + 4 * h) = rs_cur * acc;
        acc = acm_f32x4{0.f, 0.f, 0.f, 0.f};
        ++s;
        rem = s < s_end ? __builtin_amdgcn_readfirstlane(item_next.z) : 0x7fffffff;
        item = item_next;
        item_next = items[(s + 1) * 4 + g];
        rs_cur = gr.row_scale ? gr.row_scale[item.x >= 0 ? item.x : 0] : 1.f;
    };
#define ACM_STEP(Z)                                  \
    do {                                             \
        acc += (Z[0] + Z[1]) + (Z[2] + Z[3]);        \
        ACM_ISSUE(Z, q0);                            \
        q0 = q1;                                     \
        q1 = ACM_IDS(ioff);                          \
        ioff += 512;                                 \
        if (--rem == 0) finish();                    \
    } while (0)
    for (int t = sv.waves[W * 4 + 3]; t > 0; t -= 2) {
        ACM_STEP(za);
        ACM_STEP(zb);
    }
#undef ACM_STEP
#undef ACM_ISSUE
#undef ACM_IDS
}

