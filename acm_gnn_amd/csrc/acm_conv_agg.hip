// Aggregate-first form of the ACM layer (K2a forward, K3a backward) for gfx950.
//
//   P = A_low X  (narrow gather: F_in <= 16 floats per edge instead of 2 F)       -- lanes over neighbours
//   pre_L = P W_L, pre_H = (X - P) W_H, Z_I = X W_I  (3 F_in FMAs per column)      -- lanes over columns
//   ReLU / LayerNorm / sigmoid / 3x3 mix / softmax / weighted sum                  -- shared head
//
// One wave per work item; waves are persistent (grid-stride over items) so the three
// F_in x F weight panels stay in registers (lane l holds column l of every row of W).
// The backward needs no transposed SpMM when the layer input has no gradient:
//   dW_L = P^T G_L, dW_H = (X - P)^T G_H, dW_I = X^T G_I
// are outer-product reductions over rows, accumulated per lane in registers and combined
// deterministically (wave -> LDS -> per-block partial -> tree reduce).
#include "acm_conv_device.h"
#include "acm_stream_device.h"

// defined in acm_conv.hip
int acm_spmm_internal(const acm_csr* a, const void* G, int64_t ldg, int width, float* Y, int64_t ldy,
                      const acm_spmm_opts_t* o, void* workspace, size_t workspace_bytes, acm_stream_t stream,
                      bool* defer_fixup);
// defined in acm_conv_agg16.hip: the row-local forward stage in the transposed matrix-core layout (-1: not its case)
int acm_agg_epi16(const acm_conv_agg_fwd_t* p, int64_t n_rows, bool* next_done, hipStream_t s);
// ... and the row-local backward (blocks launched; 0: not its case; < 0: error)
struct GatherRole;
int acm_agg_bwd16(const acm_conv_agg_bwd_t* p, int64_t n_rows, float* partial, int max_blocks, hipStream_t s, const GatherRole* gr,
                  int gather_blocks);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
using LG = LayGrouped<4>;   // 16 lanes x 4 columns per row, 4 rows per wave (F <= 64)

template <int FP>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&v)[FP]) {
#pragma unroll
    for (int q = 0; q < FP / 4; ++q) {
        const float4 t = reinterpret_cast<const float4*>(p)[q];
        v[4 * q + 0] = t.x;
        v[4 * q + 1] = t.y;
        v[4 * q + 2] = t.z;
        v[4 * q + 3] = t.w;
    }
}

// The three F_in x F weight panels staged in LDS as [c][f][m][i] <-> W_c[f][col = m + 16 i]
// (zero beyond f_in / F), so that lane m fetches its four columns of one weight row with a
// single conflict-free ds_read_b128 (the four row-groups of a wave read the same address).
template <int FP>
__device__ __forceinline__ void stage_weights(float* wlds, const float* w_low, const float* w_high,
                                              const float* w_mlp, long ld, int f_in, int F) {
    for (int idx = threadIdx.x; idx < 3 * FP * 64; idx += 256) {
        const int c = idx / (FP * 64), f = (idx / 64) % FP, m = (idx % 64) / 4, i = idx % 4;
        const int col = m + 16 * i;
        const float* w = c == 0 ? w_low : (c == 1 ? w_high : w_mlp);
        wlds[idx] = (f < f_in && col < F) ? w[(long)f * ld + col] : 0.f;
    }
}

// pre_L = P W_L, pre_H = (x - P) W_H, z_I = x W_I for the lane's four columns.
// P and x of the group's row are parked in a per-group LDS scratch (2 FP floats) so that the loop
// over f can stay *rolled*: with a fully unrolled loop hipcc keeps all 3 FP ds_read_b128 weight rows
// in flight (96 VGPRs at FP = 8) and the row-local kernels drop to 1-2 waves/SIMD.  Each step reads
// P[f], x[f] (one address per group: LDS broadcast) and three weight rows.
template <int FP>
__device__ __forceinline__ void project(const float* wlds, float* scratch, int m, const float* __restrict__ prow,
                                        const float* __restrict__ xrow, bool active, float (&p0)[4], float (&p1)[4],
                                        float (&zi)[4]) {
    // lanes 0 .. FP/4-1 of the group fetch P, the next FP/4 fetch x (16-byte pieces); prow == nullptr: P is
    // already in the scratch (a long row whose partial sums were added by the caller)
    if (m < FP / 2 && (prow || m >= FP / 4)) {
        const float* src = (m < FP / 4) ? prow + 4 * m : xrow + 4 * (m - FP / 4);
        const float4 v = active ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(scratch + 4 * m) = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) p0[i] = p1[i] = zi[i] = 0.f;
#pragma unroll 2
    for (int f = 0; f < FP; ++f) {
        const float Pf = scratch[f], xf = scratch[FP + f];
        const float4 wl = *reinterpret_cast<const float4*>(wlds + ((0 * FP + f) * 16 + m) * 4);
        const float4 wh = *reinterpret_cast<const float4*>(wlds + ((1 * FP + f) * 16 + m) * 4);
        const float4 wm = *reinterpret_cast<const float4*>(wlds + ((2 * FP + f) * 16 + m) * 4);
        const float d = xf - Pf;
        p0[0] = fmaf(Pf, wl.x, p0[0]); p0[1] = fmaf(Pf, wl.y, p0[1]);
        p0[2] = fmaf(Pf, wl.z, p0[2]); p0[3] = fmaf(Pf, wl.w, p0[3]);
        p1[0] = fmaf(d, wh.x, p1[0]); p1[1] = fmaf(d, wh.y, p1[1]);
        p1[2] = fmaf(d, wh.z, p1[2]); p1[3] = fmaf(d, wh.w, p1[3]);
        zi[0] = fmaf(xf, wm.x, zi[0]); zi[1] = fmaf(xf, wm.y, zi[1]);
        zi[2] = fmaf(xf, wm.z, zi[2]); zi[3] = fmaf(xf, wm.w, zi[3]);
    }
}

// P (uniform in the 16-lane group) -> projections -> head -> out / att for one row.
template <int FP, int K, bool FULL = false>
__device__ __forceinline__ void agg_fwd_row(const acm_conv_agg_fwd_t& p, const float* wlds, const float* hlds,
                                            float* scratch, const float* mixm, int row, int lane, const CsrView& csr,
                                            const float* __restrict__ partial, const AcmDropCtx& dc,
                                            bool p_in_scratch = false) {
    constexpr bool active = true;
    const int F = FULL ? 64 : p.f_out, m = lane & 15;     // FULL: f_out == 64, the column guards fold away
    float H[K][4];
    {
        // A long row's work items left partial sums of P in the slots: the 16 lanes of the group add them here (slot
        // order within a lane, fixed DPP tree across lanes) instead of a separate fix-up launch after the gather.
        const float* prow = p_in_scratch ? nullptr : p.agg + (long)row * p.ld_agg;
        if (K == 3 && partial && !p_in_scratch) {
            const int li = csr.long_index[row];                  // uniform in the group
            if (li >= 0) {
                const AcmLongRow lr = csr.long_rows[li];
                float a[FP];
#pragma unroll
                for (int f = 0; f < FP; ++f) a[f] = 0.f;
                for (int s = lr.slot_begin + m; s < lr.slot_end; s += 16) {
                    float v[FP];
                    load_vec<FP>(partial + (long)s * FP, v);
#pragma unroll
                    for (int f = 0; f < FP; ++f) a[f] += v[f];
                }
                const float rs = p.row_scale ? p.row_scale[row] : 1.f;
#pragma unroll
                for (int f = 0; f < FP; ++f) a[f] = rs * acm_group_sum<16>(a[f]);
                if (m == 0) {
#pragma unroll
                    for (int f = 0; f < FP; ++f) {
                        scratch[f] = a[f];
                        p.agg[(long)row * p.ld_agg + f] = a[f];      // saved for the backward
                    }
                }
                prow = nullptr;
            }
        }
        float p0[4], p1[4], zi[4];
        project<FP>(wlds, scratch, m, prow, p.xs + (long)row * p.ld_xs, true, p0, p1, zi);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = m + 16 * i < F;
            H[0][i] = ok ? (p.relu_after ? fmaxf(p0[i], 0.f) : p0[i]) : 0.f;
            H[1][i] = ok ? (p.relu_after ? fmaxf(p1[i], 0.f) : p1[i]) : 0.f;
            H[2][i] = ok ? (p.relu_mlp ? fmaxf(zi[i], 0.f) : zi[i]) : 0.f;
        }
    }
    if (K == 4) {                                  // structure channel: relu(deg * (A_low S) - S)
        const float dg = p.deg[row];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cc = (m + 16 * i < F) ? m + 16 * i : 0;
            const float v = dg * p.ps[(unsigned)row * (unsigned)p.ld_ps + cc] - p.ss[(unsigned)row * (unsigned)p.ld_ss + cc];
            H[K - 1][i] = (m + 16 * i < F) ? fmaxf(v, 0.f) : 0.f;
        }
    }
    RowHead<K> rh;
    row_head<K>(hlds, mixm, acm_opaque(m), F, p.layernorm != 0, H, rh);
    if (p.head_stats && m == 0 && active) row_head_store<K>(p.head_stats + (long)row * p.ld_head_stats, rh);
    float df[4];
    acm_drop4(dc, row, m, df);        // dc: read once per launch (its step counter is a global load)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = m + 16 * i;
        if (col < F && active) {
            float o = rh.alpha[0] * H[0][i] + rh.alpha[1] * H[1][i] + rh.alpha[2] * H[2][i];
            if (K == 4) o = fmaf(rh.alpha[K - 1], H[K - 1][i], o);
            o *= p.scale;
            if (p.post_relu) o = fmaxf(o, 0.f);
            if (p.post_scale) o *= p.post_scale[(long)row * p.ld_post_scale + col];
            if (p.post_drop.p > 0.f) o *= df[i];
            p.out[(long)row * p.ld_out + col] = o;
        }
    }
    if (m == 0 && active)
        *reinterpret_cast<float4*>(p.att + (long)row * 4) =
            make_float4(rh.alpha[0], rh.alpha[1], rh.alpha[2], K == 4 ? rh.alpha[K - 1] : 0.f);
}

// Forward = two launches: (1) P = A_low X through the lean narrow-gather kernel of acm_spmm (few
// registers => 8 waves/SIMD in flight, which is what a request-latency-bound gather needs; the
// fused version held the epilogue's 156 VGPRs during the gather and ran at 3 waves/SIMD), (2) this
// streaming row-local kernel: 4 rows per wave, 16 lanes x 4 columns each.
template <int FP, int K>
__global__ __launch_bounds__(256) void agg_epilogue_kernel(acm_conv_agg_fwd_t p, int n_rows, CsrView csr,
                                                           const float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float wlds[3 * FP * 64 + 3 * K * 64 + 16 * 2 * FP];
    float* hlds = wlds + 3 * FP * 64;
    float* scratch = hlds + 3 * K * 64 + (threadIdx.x >> 4) * 2 * FP;      // this 16-lane group's P | x
    stage_weights<FP>(wlds, p.w_low, p.w_high, p.w_mlp, p.ld_w, p.f_in, p.f_out);
    stage_head_params<K>(hlds, p.att_vec, p.ln_weight, p.ln_bias, p.layernorm, p.f_out);
    __syncthreads();
    float mixm[K * K];
#pragma unroll
    for (int q = 0; q < K * K; ++q) mixm[q] = p.att_mix[q];
    const AcmDropCtx dc = acm_drop_ctx(p.post_drop);
    const int lane = threadIdx.x & 63;
    const int m = lane & 15;
    for (int row = blockIdx.x * 16 + (threadIdx.x >> 4); row < n_rows; row += gridDim.x * 16) {
        agg_fwd_row<FP, K>(p, wlds, hlds, scratch, mixm, row, lane, csr, partial, dc);
        if (p.agg_copy && m < FP / 2) {           // the P | x rows just read (still parked in the group's scratch)
            const float4 v = *reinterpret_cast<const float4*>(scratch + 4 * m);
            float* dst = (m < FP / 4) ? p.agg_copy + (long)row * p.ld_agg_copy + 4 * m
                                      : p.xs_copy + (long)row * p.ld_xs_copy + 4 * (m - FP / 4);
            *reinterpret_cast<float4*>(dst) = v;
        }
    }
}

// The gather and the epilogue in ONE kernel (three channels, f_pad <= 8, 16 lanes per work item): the narrow gather's
// 16-lane group ends with the aggregated row P replicated in its lanes -- exactly the input layout of agg_fwd_row (16
// lanes x 4 columns of one row) -- so the projections and the head run right there, on lanes that would otherwise
// idle through the next item's gather latency.  Same software pipeline over the work list as spmm_narrow_kernel.
// The pieces of a long row fill whole windows of the work list (acm_csr.cpp, build_items): a workgroup round is one
// window, the pieces meet in LDS and the first piece's group finishes the row -- no partial slots, no second launch.
template <int FP, bool FULL>
__global__ __launch_bounds__(256) void agg_fused_kernel(acm_conv_agg_fwd_t p, CsrView csr) {
    constexpr int K = 3, GS = 16, GPB = 16;
    static_assert(GPB == ACM_WINDOW, "one window of work items per workgroup round");
    __shared__ __attribute__((aligned(16))) float wlds[3 * FP * 64 + 3 * K * 64 + 16 * 2 * FP];
    __shared__ float coop[ACM_WINDOW * FP];
    float* hlds = wlds + 3 * FP * 64;
    float* scratch = hlds + 3 * K * 64 + (threadIdx.x >> 4) * 2 * FP;
    stage_weights<FP>(wlds, p.w_low, p.w_high, p.w_mlp, p.ld_w, p.f_in, p.f_out);
    stage_head_params<K>(hlds, p.att_vec, p.ln_weight, p.ln_bias, p.layernorm, p.f_out);
    __syncthreads();
    float mixm[K * K];
#pragma unroll
    for (int q = 0; q < K * K; ++q) mixm[q] = p.att_mix[q];
    const AcmDropCtx dc = acm_drop_ctx(p.post_drop);
    const int gl = threadIdx.x & 15, lane = threadIdx.x & 63;
    const int G = gridDim.x * GPB;
    int w = blockIdx.x * GPB + (threadIdx.x >> 4);
    if (w >= csr.n_items) return;
    const bool unit = csr.vals == nullptr;
    AcmItem it = csr.items[w];
    int k0 = it.begin;
    bool va = k0 + gl < it.end, vb = k0 + gl + GS < it.end;
    int ja = va ? csr.indices[k0 + gl] : 0, jb = vb ? csr.indices[k0 + gl + GS] : 0;
    float aa = va ? (unit ? 1.f : csr.vals[k0 + gl]) : 0.f, ab = vb ? (unit ? 1.f : csr.vals[k0 + gl + GS]) : 0.f;
    while (true) {
        const int wn = w + G;
        const bool has_next = wn < csr.n_items;
        AcmItem itn = it;
        if (has_next) itn = csr.items[wn];
        float acc[FP];
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[f] = 0.f;
        while (true) {
            float za[FP], zb[FP];
            load_vec<FP>(p.xg + (long)ja * p.ld_xg, za);
            load_vec<FP>(p.xg + (long)jb * p.ld_xg, zb);
            const int k1 = k0 + 2 * GS;
            const bool more = k1 < it.end;
            const int pb = more ? k1 : itn.begin;
            const int pe = more ? it.end : (has_next ? itn.end : pb);
            const bool pva = pb + gl < pe, pvb = pb + gl + GS < pe;
            const int nja = pva ? csr.indices[pb + gl] : 0, njb = pvb ? csr.indices[pb + gl + GS] : 0;
            const float naa = pva ? (unit ? 1.f : csr.vals[pb + gl]) : 0.f, nab = pvb ? (unit ? 1.f : csr.vals[pb + gl + GS]) : 0.f;
#pragma unroll
            for (int f = 0; f < FP; ++f) {
                acc[f] = va ? fmaf(aa, za[f], acc[f]) : acc[f];
                acc[f] = vb ? fmaf(ab, zb[f], acc[f]) : acc[f];
            }
            ja = nja, jb = njb, aa = naa, ab = nab, va = pva, vb = pvb;
            if (!more) break;
            k0 = k1;
        }
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[f] = acm_group_sum<GS>(acc[f]);
        bool finish = it.slot < 0;
        if (w / ACM_WINDOW < csr.n_windows) {                   // a window of pieces (uniform over the workgroup):
            const int g = threadIdx.x >> 4;                     // they meet in LDS, the first piece finishes the row
            if (gl == 0) {
#pragma unroll
                for (int f = 0; f < FP; ++f) coop[g * FP + f] = acc[f];
            }
            __syncthreads();
            const AcmLongRow lr = csr.long_rows[csr.long_index[it.row]];
            finish = it.slot == lr.slot_begin;
            if (finish) {
                const int pieces = lr.slot_end - lr.slot_begin;
#pragma unroll
                for (int f = 0; f < FP; ++f) {
                    float t = 0.f;
                    for (int q = 0; q < pieces; ++q) t += coop[(g + q) * FP + f];
                    acc[f] = t;
                }
            }
            __syncthreads();
        }
        if (finish) {
            const float rs = p.row_scale ? p.row_scale[it.row] : 1.f;
            if (gl == 0) {
#pragma unroll
                for (int f = 0; f < FP; ++f) {
                    const float v = rs * acc[f];
                    scratch[f] = v;
                    p.agg[(long)it.row * p.ld_agg + f] = v;          // P = A_low X, saved for the backward
                }
            }
            agg_fwd_row<FP, K, FULL>(p, wlds, hlds, scratch, mixm, it.row, lane, csr, nullptr, dc, true);
        }
        if (!has_next) break;
        it = itn;
        w = wn;
        k0 = it.begin;
    }
}

// f_pad = 8 (32-byte rows): the same kernel with TWO lanes per neighbour, each fetching one 16-byte half of the row.
// A lane can load at most 16 bytes per instruction, so with one neighbour per lane a row costs two requests from every
// lane, each to a distinct cache line (128 line look-ups per 64 neighbours); with adjacent lanes on the two halves of
// one row the address coalescer sees 32 lines per 32 neighbours -- half the look-ups per byte, which is what bounds the
// gather once the rows hit in L1/L2 (scripts/probe_gather.py: 63 us with every row in L1).
// Lane (e = gl >> 1, h = gl & 1) of the 16-lane group: neighbours k0 + e + 8 u (u = 0..3), columns 4 h .. 4 h + 3.
template <bool FULL>
__device__ __forceinline__ void agg_fused_pair_body(const acm_conv_agg_fwd_t& p, const CsrView& csr) {
    constexpr int FP = 8, K = 3, GPB = 16, U = 4, STEP = 8 * U;
    static_assert(GPB == ACM_WINDOW, "one window of work items per workgroup round");
    __shared__ __attribute__((aligned(16))) float wlds[3 * FP * 64 + 3 * K * 64 + 16 * 2 * FP];
    __shared__ float coop[ACM_WINDOW * FP];
    float* hlds = wlds + 3 * FP * 64;
    float* scratch = hlds + 3 * K * 64 + (threadIdx.x >> 4) * 2 * FP;
    stage_weights<FP>(wlds, p.w_low, p.w_high, p.w_mlp, p.ld_w, p.f_in, p.f_out);
    stage_head_params<K>(hlds, p.att_vec, p.ln_weight, p.ln_bias, p.layernorm, p.f_out);
    __syncthreads();
    float mixm[K * K];
#pragma unroll
    for (int q = 0; q < K * K; ++q) mixm[q] = p.att_mix[q];
    const AcmDropCtx dc = acm_drop_ctx(p.post_drop);
    const int gl = threadIdx.x & 15, lane = threadIdx.x & 63, e = gl >> 1, h = gl & 1;
    const int G = gridDim.x * GPB;
    int w = blockIdx.x * GPB + (threadIdx.x >> 4);
    if (w >= csr.n_items) return;
    const bool unit = csr.vals == nullptr;
    const float* xh = p.xg + 4 * h;
    AcmItem it = csr.items[w];
    int k0 = it.begin;
    int j[U];
    float a[U];
    bool v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = k0 + e + 8 * u;
        v[u] = k < it.end;
        j[u] = v[u] ? csr.indices[k] : 0;
        a[u] = v[u] ? (unit ? 1.f : csr.vals[k]) : 0.f;
    }
    while (true) {
        const int wn = w + G;
        const bool has_next = wn < csr.n_items;
        AcmItem itn = it;
        if (has_next) itn = csr.items[wn];
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        while (true) {
            float4 z[U];
#pragma unroll
            for (int u = 0; u < U; ++u) z[u] = *reinterpret_cast<const float4*>(xh + (long)j[u] * p.ld_xg);
            const int k1 = k0 + STEP;
            const bool more = k1 < it.end;
            const int pb = more ? k1 : itn.begin;
            const int pe = more ? it.end : (has_next ? itn.end : pb);
            int nj[U];
            float na[U];
            bool nv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = pb + e + 8 * u;
                nv[u] = k < pe;
                nj[u] = nv[u] ? csr.indices[k] : 0;
                na[u] = nv[u] ? (unit ? 1.f : csr.vals[k]) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc[0] = v[u] ? fmaf(a[u], z[u].x, acc[0]) : acc[0];
                acc[1] = v[u] ? fmaf(a[u], z[u].y, acc[1]) : acc[1];
                acc[2] = v[u] ? fmaf(a[u], z[u].z, acc[2]) : acc[2];
                acc[3] = v[u] ? fmaf(a[u], z[u].w, acc[3]) : acc[3];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) j[u] = nj[u], a[u] = na[u], v[u] = nv[u];
            if (!more) break;
            k0 = k1;
        }
        // sum over the eight lanes of the group with the same half (lanes gl, gl^2, gl+-4, gl+-8): fixed order
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] += acm_dpp<0x4E>(acc[i]);     // quad_perm [2,3,0,1]
            acc[i] += acm_dpp<0x124>(acc[i]);    // row_ror:4
            acc[i] += acm_dpp<0x128>(acc[i]);    // row_ror:8
        }
        bool finish = it.slot < 0;
        if (w / ACM_WINDOW < csr.n_windows) {                   // a window of pieces: see agg_fused_kernel
            const int g = threadIdx.x >> 4;
            if (gl < 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) coop[g * FP + 4 * h + i] = acc[i];
            }
            __syncthreads();
            const AcmLongRow lr = csr.long_rows[csr.long_index[it.row]];
            finish = it.slot == lr.slot_begin;
            if (finish && gl < 2) {
                const int pieces = lr.slot_end - lr.slot_begin;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = 0.f;
                    for (int q = 0; q < pieces; ++q) t += coop[(g + q) * FP + 4 * h + i];
                    acc[i] = t;
                }
            }
            __syncthreads();
        }
        if (finish) {
            const float rs = p.row_scale ? p.row_scale[it.row] : 1.f;
            if (gl < 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float val = rs * acc[i];
                    scratch[4 * h + i] = val;
                    p.agg[(long)it.row * p.ld_agg + 4 * h + i] = val;
                }
            }
            agg_fwd_row<FP, K, FULL>(p, wlds, hlds, scratch, mixm, it.row, lane, csr, nullptr, dc, true);
        }
        if (!has_next) break;
        it = itn;
        w = wn;
        k0 = it.begin;
    }
}


template <bool FULL>
__global__ __launch_bounds__(256) void agg_fused_pair_kernel(acm_conv_agg_fwd_t p, CsrView csr) {
    agg_fused_pair_body<FULL>(p, csr);
}

// ---------------------------------------------------------------- backward
// flat parameter-gradient vector: [dW_low f_in*F][dW_high][dW_mlp][dv 3F][dgamma 3F][dbeta 3F][dmix 9]
//
// dW_c = A_c^T G_c with A_L = P, A_H = X - P, A_I = X is accumulated on the matrix pipe: each
// wave step covers 4 rows = the K dimension of v_mfma_f32_16x16x4_f32.  Lane (g, m) supplies
//   A[i = m][k = g] = A_c[row_g][f = m]      (one per-lane load of P / X element m)
//   B[k = g][j = m] = G_c[row_g][16 t + m]   (exactly the LayGrouped column the lane owns)
// and the accumulator tile t holds dW_c[f = 4 (lane >> 4) + r][16 t + (lane & 15)].
//
// Register discipline (the kernel is VALU-bound and lives or dies by waves/SIMD): the row is
// processed in two passes.  Pass 1 keeps only H (12 values) and nine scalars (mean, rstd, g per
// channel, then alpha / ds); pass 2 walks the channels one at a time, recomputes xhat from
// (H, mean, rstd), and hands each channel's G straight to the MFMAs.  att_vec / LayerNorm
// gamma, beta sit in LDS next to the weights ([array][c][m][i], one ds_read_b128 per use).
// FULL: f_out == 64 (the reference's hidden width), every column guard `m + 16 i < F` folds away at compile time.
template <int FP, int K, bool FULL>
__device__ __forceinline__ void agg_bwd_body(const acm_conv_agg_bwd_t& p, int n_rows, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    constexpr int BW = 4;                         // waves per workgroup
    const int F = FULL ? 64 : p.f_out, f_in = p.f_in;
    const int npg = 3 * f_in * F + 3 * K * F + K * K;
    float* wlds = lds;                           // 3 * FP * 64 floats
    float* hlds = lds + 3 * FP * 64;             // 3 * K * 64 floats
    float* scratch = hlds + 3 * K * 64 + (threadIdx.x >> 4) * 2 * FP;   // per-group P | x; all dead after the row loop
    stage_weights<FP>(wlds, p.w_low, p.w_high, p.w_mlp, p.ld_w, f_in, F);
    stage_head_params<K>(hlds, p.att_vec, p.ln_weight, p.ln_bias, p.layernorm, F);
    __syncthreads();
    f32x4 acc[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float pA[K][4], pS[K], dmix1 = 0.f;          // head-parameter accumulators (see row_channel_backward)
    const int qc = (m < K * K ? m : 0) / K, qj = (m < K * K ? m : 0) % K;    // the att_mix element this lane accumulates
#pragma unroll
    for (int c = 0; c < K; ++c) {
        pS[c] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) pA[c][i] = 0.f;
    }
    float mixm[K * K];
#pragma unroll
    for (int q = 0; q < K * K; ++q) mixm[q] = p.att_mix[q];
    const bool ln = p.layernorm != 0;
    const bool out_mask = p.out != nullptr && p.post_relu && !p.post_scale;
    const float post_gain = p.post_drop.p > 0.f ? acm_drop_ctx(p.post_drop).inv_keep : 1.f;

    for (int r0 = (blockIdx.x * BW + wv) * 4; r0 < n_rows; r0 += gridDim.x * BW * 4) {
        const int row = r0 + g;
        const bool active = row < n_rows;
        const long rr = active ? row : 0;
        float H[K][4], dO[4];
        if (K == 4) {
            const float dg0 = active ? p.deg[rr] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = active && m + 16 * i < F;
                const int cc = ok ? m + 16 * i : 0;
                const float v = dg0 * p.ps[(unsigned)rr * (unsigned)p.ld_ps + cc] - p.ss[(unsigned)rr * (unsigned)p.ld_ss + cc];
                H[K - 1][i] = ok ? fmaxf(v, 0.f) : 0.f;
            }
        }
        {
            float p0[4], p1[4], zi[4];
            // 32-bit element offsets (the host checks n_rows * ld < 2^31): a long x long product is three quarter-rate multiplies
            project<FP>(wlds, scratch, m, p.agg + (unsigned)rr * (unsigned)p.ld_agg, p.xs + (unsigned)rr * (unsigned)p.ld_xs, active, p0, p1, zi);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = active && m + 16 * i < F;
                H[0][i] = ok ? (p.relu_after ? fmaxf(p0[i], 0.f) : p0[i]) : 0.f;
                H[1][i] = ok ? (p.relu_after ? fmaxf(p1[i], 0.f) : p1[i]) : 0.f;
                H[2][i] = ok ? (p.relu_mlp ? fmaxf(zi[i], 0.f) : zi[i]) : 0.f;
                const float go = p.grad_out[(unsigned)rr * (unsigned)p.ld_grad_out + (ok ? m + 16 * i : 0)];
                dO[i] = ok ? go : 0.f;
            }
        }
        // ---- pass 1: per-channel statistics and the attention scalars
        const int mm = acm_opaque(m);
        RowHead<K> rh;
        if (p.head_stats) row_head_load<K>(p.head_stats + (unsigned)rr * (unsigned)p.ld_head_stats, rh);   // as the forward computed them
        else row_head<K>(hlds, mixm, mm, F, ln, H, rh);
        if (out_mask) {                           // the forward's output tells which elements the post-op let through
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float o = p.out[(unsigned)rr * (unsigned)p.ld_out + ((m + 16 * i < F) ? m + 16 * i : 0)];
                dO[i] = (o != 0.f) ? dO[i] * post_gain : 0.f;
            }
        } else {
            row_post_backward<K>(p, rh, H, active, rr, m, F, dO);
        }
        float ds[K];
        row_head_backward_scalars<K>(rh, mixm, p.scale, H, dO, ds, qc, qj, dmix1);
        // ---- pass 2: one channel at a time -> G_c -> MFMA
        const float Pm = (m < FP) ? scratch[m] : 0.f;            // zero for inactive rows (project() stored zeros)
        const float xm = (m < FP) ? scratch[FP + m] : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool relu_c = (c < 2) ? (p.relu_after != 0) : (p.relu_mlp != 0);
            const float aop = (c == 0) ? Pm : (c == 1 ? xm - Pm : xm);
            float G[4];
            row_channel_backward<K>(hlds, c, mm, F, ln, p.scale, rh, ds[c], H[c], dO, pA[c], pS[c], G);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool keep = active && (m + 16 * t < F) && (!relu_c || H[c][t] > 0.f);
                acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, keep ? G[t] : 0.f, acc[c][t], 0, 0, 0);
            }
        }
        if (K == 4) {                              // structure channel: deg * G_S goes to memory for A_low^T
            float G[4];
            row_channel_backward<K>(hlds, K - 1, mm, F, ln, p.scale, rh, ds[K - 1], H[K - 1], dO, pA[K - 1], pS[K - 1], G);
            const float dg1 = (active && p.g_struc_scale) ? p.g_struc_scale[rr] : 1.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (active && m + 16 * t < F)
                    p.g_struc[(unsigned)rr * (unsigned)p.ld_g_struc + m + 16 * t] = H[K - 1][t] > 0.f ? dg1 * G[t] : 0.f;
        }
    }
    // head-parameter partials: combine the four row-groups of the wave
    float dv[K][4], dgam[K][4], dbet[K][4];
#pragma unroll
    for (int c = 0; c < K; ++c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pA[c][i] = acm_cross_row_sum(pA[c][i]);
        pS[c] = acm_cross_row_sum(pS[c]);
        row_param_grads<K>(hlds, c, m, pA[c], pS[c], dv[c], dgam[c], dbet[c]);      // hlds is still intact here
    }
    dmix1 = acm_cross_row_sum(dmix1);
    __syncthreads();                              // every wave is done with wlds / hlds
    float* slab = lds + wv * npg;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 4 * g + r, col = 16 * t + m;
                if (f < f_in && col < F) slab[(c * f_in + f) * F + col] = acc[c][t][r];
            }
    if (g == 0) {
        const int base = 3 * f_in * F;
#pragma unroll
        for (int c = 0; c < K; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = m + 16 * i;
                if (col < F) {
                    slab[base + (0 * K + c) * F + col] = dv[c][i];
                    slab[base + (1 * K + c) * F + col] = dgam[c][i];
                    slab[base + (2 * K + c) * F + col] = dbet[c][i];
                }
            }
    }
    if (g == 0 && m < K * K) slab[3 * f_in * F + 3 * K * F + m] = dmix1;
    __syncthreads();
    // the slab is stored in groups of 32 parameters, partial[q / 32][block][q % 32] (acm_reduce_seg_t.elem_stride): whole
    // 128-byte lines here, and the second phase reads one line per block and group instead of one float per line
    for (int q = threadIdx.x; q < npg; q += BW * 64) {
        const float v = (lds[q] + lds[npg + q]) + (lds[2 * npg + q] + lds[3 * npg + q]);
        partial[((long)(q >> 5) * gridDim.x + blockIdx.x) * 32 + (q & 31)] = v;
    }
}

template <int FP, int K, bool FULL>
__global__ __launch_bounds__(256, 2) void agg_bwd_kernel(acm_conv_agg_bwd_t p, int n_rows, float* __restrict__ partial) {
    agg_bwd_body<FP, K, FULL>(p, n_rows, partial);
}
int agg_pad(int f_in) { return f_in <= 4 ? 4 : (f_in <= 8 ? 8 : 16); }

// Persistent grid of agg_bwd_kernel: as many workgroups as stay resident.  Three channels: 151 VGPRs, three workgroups
// (3 waves/SIMD) per CU -- 768 blocks take the twitch layer 85 -> 77 us against 512 (1 024 do not fit: 82 us);
// with the structure channel (185-191 VGPRs) two fit.
int agg_bwd_blocks(int64_t n_rows, int n_channels, size_t lds_bytes) {
    int64_t nb = (n_rows + 15) / 16;
    const int cap = (n_channels == 3 && 3 * lds_bytes <= 160 * 1024) ? 768 : 512;     // registers and LDS of three
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

template <class P>
int check_common(const P* p, const char* who) {
    ACM_REQUIRE(p->f_in >= 1 && p->f_in <= 16, ACM_EUNSUPPORTED, "%s: f_in %d outside 1..16", who, p->f_in);
    ACM_REQUIRE(p->f_out >= 1 && p->f_out <= 64, ACM_EUNSUPPORTED, "%s: f_out %d outside 1..64", who, p->f_out);
    ACM_REQUIRE(p->f_pad == agg_pad(p->f_in), ACM_ESHAPE, "%s: f_pad %d, expected %d for f_in %d", who, p->f_pad,
                agg_pad(p->f_in), p->f_in);
    ACM_REQUIRE(p->w_low && p->w_high && p->w_mlp && p->att_mix && p->xs, ACM_EINVAL, "%s: NULL pointer", who);
    ACM_REQUIRE(p->ld_w >= p->f_out, ACM_ESHAPE, "%s: ld_w too small", who);
    ACM_REQUIRE(((uintptr_t)p->xs) % 16 == 0 && (p->ld_xs * 4) % 16 == 0 && p->ld_xs >= p->f_pad, ACM_EINVAL,
                "%s: xs rows must be 16-byte aligned and f_pad long", who);
    ACM_REQUIRE(p->n_channels == 3 || p->n_channels == 4, ACM_ESHAPE, "%s: n_channels %d", who, p->n_channels);
    ACM_REQUIRE(!p->head_stats || (((uintptr_t)p->head_stats) % 16 == 0 && p->ld_head_stats % 4 == 0 &&
                                   p->ld_head_stats >= 4 * p->n_channels), ACM_EINVAL,
                "%s: head_stats rows must be 16-byte aligned and 4 * n_channels long", who);
    for (int c = 0; c < p->n_channels; ++c) {
        ACM_REQUIRE(p->att_vec[c], ACM_EINVAL, "%s: att_vec[%d] NULL", who, c);
        ACM_REQUIRE(!p->layernorm || (p->ln_weight[c] && p->ln_bias[c]), ACM_EINVAL, "%s: LayerNorm pointers NULL", who);
    }
    if (p->n_channels == 4)
        ACM_REQUIRE(p->ps && p->ss && p->deg && p->ld_ps >= p->f_out && p->ld_ss >= p->f_out, ACM_EINVAL,
                    "%s: structure-channel pointers NULL / leading dimensions too small", who);
    return ACM_OK;
}

StreamView stream_view(const AcmStreams* t) {
    StreamView sv;
    sv.ids = t->ids;
    sv.waves = t->waves;
    sv.items = t->items;
    sv.long_rows = t->long_rows;
    sv.long_index = t->long_index;
    sv.counters = t->counters;
    sv.slots = t->slots;
    sv.ids_bytes = (unsigned)((t->total_steps + ACM_STREAM_PAD_STEPS) * 512);
    sv.slots_bytes = (unsigned)(t->n_slots * 32);
    sv.n_waves = t->n_waves;
    return sv;
}

}  // namespace

// The requested next-layer projection for the paths whose epilogue does not carry it: a launch of its own.
static int next_projection(const acm_csr_t* a, const acm_conv_agg_fwd_t* p, acm_stream_t stream) {
    if (p->next_f <= 0) return ACM_OK;
    return acm_proj_fwd(a->n_rows, p->f_out, p->next_f, p->out, p->ld_out, p->next_w_low, p->next_w_high, p->next_w_mlp,
                        p->next_ld_w, p->next_relu, p->next_zlh, p->ld_next_zlh, p->next_zi, p->ld_next_zi, stream);
}

extern "C" int acm_conv_agg_fwd(const acm_csr_t* a, const acm_conv_agg_fwd_t* p, void* workspace,
                                size_t workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(a && p, ACM_EINVAL, "acm_conv_agg_fwd: NULL argument");
    int st = check_common(p, "acm_conv_agg_fwd");
    if (st != ACM_OK) return st;
    ACM_REQUIRE((p->xg || p->agg_given) && p->out && p->agg && p->att, ACM_EINVAL, "acm_conv_agg_fwd: NULL tensor pointer");
    ACM_REQUIRE((p->agg_given || (((uintptr_t)p->xg) % 16 == 0 && (p->ld_xg * 4) % 16 == 0 && p->ld_xg >= p->f_pad)) &&
                    ((uintptr_t)p->agg) % 16 == 0 && (p->ld_agg * 4) % 16 == 0 && p->ld_agg >= p->f_pad &&
                    ((uintptr_t)p->att) % 16 == 0, ACM_EINVAL,
                "acm_conv_agg_fwd: xg / agg rows must be 16-byte aligned and f_pad long");
    if (p->n_channels == 4) {       // agg_fwd_row indexes ps / ss with 32-bit element offsets
        const int64_t ld_max = p->ld_ps > p->ld_ss ? p->ld_ps : p->ld_ss;
        ACM_REQUIRE(a->n_rows * ld_max < (int64_t)INT32_MAX, ACM_EUNSUPPORTED,
                    "acm_conv_agg_fwd: too many rows for 32-bit offsets into ps / ss");
    }
    if (p->next_f != 0) {
        ACM_REQUIRE(p->next_f >= 1 && p->next_f <= 2, ACM_EUNSUPPORTED, "acm_conv_agg_fwd: next_f %d (fused next projection: 1 or 2)", p->next_f);
        ACM_REQUIRE(p->next_w_low && p->next_w_high && p->next_w_mlp && p->next_zlh && p->next_zi && p->next_ld_w >= p->next_f &&
                        p->ld_next_zlh >= 2 * p->next_f && p->ld_next_zi >= p->next_f, ACM_EINVAL,
                    "acm_conv_agg_fwd: fused next projection: NULL pointer / leading dimension too small");
    }
    if (a->n_rows == 0) return ACM_OK;
    hipStream_t s = (hipStream_t)stream;
    bool next_done = false;
    ACM_REQUIRE(!p->agg_copy || (p->agg_given && p->xs_copy && ((uintptr_t)p->agg_copy) % 16 == 0 && ((uintptr_t)p->xs_copy) % 16 == 0 &&
                                 (p->ld_agg_copy * 4) % 16 == 0 && (p->ld_xs_copy * 4) % 16 == 0 && p->ld_agg_copy >= p->f_pad &&
                                 p->ld_xs_copy >= p->f_pad && p->agg_copy != p->agg && p->xs_copy != p->xs), ACM_EINVAL,
                "acm_conv_agg_fwd: agg_copy / xs_copy need agg_given, 16-byte aligned rows of f_pad floats, no aliasing");
    ACM_REQUIRE(!p->next_x || (p->agg_copy && p->ld_next_x >= p->f_in && p->next_drop.p >= 0.f && p->next_drop.p < 1.f &&
                               (p->next_drop.p == 0.f || p->next_drop.step)), ACM_EINVAL,
                "acm_conv_agg_fwd: next_x needs agg_copy / xs_copy (xs is refilled in place), ld_next_x >= f_in and a valid next_drop");
    // (0) fused form: gather + epilogue in one kernel (long rows included)
    if (!p->agg_given) {
        const double avg = (double)a->nnz / (double)(a->n_rows > 0 ? a->n_rows : 1);
        const bool fused = acm_tuning().agg_fused != 0 && p->n_channels == 3 && p->f_pad <= 8 && avg > 12.0 && avg <= 160.0 &&
                           ((uintptr_t)p->xg) % 16 == 0 && (p->ld_xg * sizeof(float)) % 16 == 0 &&
                           (a->n_long == 0 || a->long_index != nullptr) &&
                           a->n_multi == 0;       // (rows of several windows: the two-stage form, whose gather finishes them)
        if (fused) {
            const CsrView cv = acm_view(a);
            int grid = (int)((a->n_items + 15) / 16);
            // every block stages the weights and head parameters (8.4 KB) before it starts: 12 blocks per CU keep that
            // prologue small against the gather (8192 blocks: 125 us, 3072: 115 us, 1024: 125 us on the twitch graph)
            if (grid > 3072) grid = 3072;
            const bool full = p->f_out == 64;
            if (p->f_pad == 4) {
                if (full) hipLaunchKernelGGL((agg_fused_kernel<4, true>), dim3(grid), dim3(256), 0, s, *p, cv);
                else hipLaunchKernelGGL((agg_fused_kernel<4, false>), dim3(grid), dim3(256), 0, s, *p, cv);
            } else {                                  // 32-byte rows: two lanes per neighbour
                if (full) hipLaunchKernelGGL((agg_fused_pair_kernel<true>), dim3(grid), dim3(256), 0, s, *p, cv);
                else hipLaunchKernelGGL((agg_fused_pair_kernel<false>), dim3(grid), dim3(256), 0, s, *p, cv);
            }
            ACM_CHECK_HIP(hipGetLastError());
            return next_projection(a, p, stream);
        }
    }
    // (1) P = A_low X  -> p->agg  (also the tensor saved for the backward); row_scale for a pattern-only a_low
    acm_spmm_opts_t o = {nullptr, p->row_scale, nullptr, 0, nullptr, 0, 0};
    // three channels: the long rows' partial sums stay in the workspace and the epilogue kernel adds them (one launch
    // less); with the structure channel the second gather reuses the workspace, so the fix-up runs right away
    // (the sixteen-rows-per-wave stage of acm_conv_agg16.hip reads finished rows of P: the fix-up runs right away there)
    const bool epi16 = p->f_out == 64 && (acm_tuning().rows16 & ACM_ROWS16_EPI) != 0;
    bool defer = !p->agg_given && p->n_channels == 3 && a->n_long > 0 && a->long_index != nullptr && !epi16;
    if (!p->agg_given) {
        st = acm_spmm_internal(a, p->xg, p->ld_xg, p->f_pad, p->agg, p->ld_agg, &o, workspace, workspace_bytes, stream, &defer);
        if (st != ACM_OK) return st;
    }
    // (1b) structure channel: PS = A_low S -> p->ps (F wide; bf16 operand optional)
    if (p->n_channels == 4) {
        ACM_REQUIRE(p->sg, ACM_EINVAL, "acm_conv_agg_fwd: sg is NULL");
        o.g_bf16 = p->sg_bf16;
        st = acm_spmm_ex(a, p->sg, p->ld_sg, p->f_out, p->ps, p->ld_ps, &o, workspace, workspace_bytes, stream);
        if (st != ACM_OK) return st;
    }
    // (2) projections + head, row-local
    if (epi16 && !defer) {
        st = acm_agg_epi16(p, a->n_rows, &next_done, s);        // (carries next_x: the refill of xs)
        if (st == ACM_OK) return next_done ? ACM_OK : next_projection(a, p, stream);
        if (st > 0) return st;
    }
    int grid = (int)((a->n_rows + 15) / 16);
    if (grid > 2048) grid = 2048;
    const CsrView cv = acm_view(a);
#define ACM_EPI(FPv)                                                                                           \
    do {                                                                                                       \
        if (p->n_channels == 3)                                                                                \
            hipLaunchKernelGGL((agg_epilogue_kernel<FPv, 3>), dim3(grid), dim3(256), 0, s, *p, (int)a->n_rows, cv, \
                               defer ? (const float*)workspace : nullptr);                                     \
        else                                                                                                   \
            hipLaunchKernelGGL((agg_epilogue_kernel<FPv, 4>), dim3(grid), dim3(256), 0, s, *p, (int)a->n_rows, cv, \
                               (const float*)nullptr);                                                         \
    } while (0)
    if (p->f_pad == 4) ACM_EPI(4);
    else if (p->f_pad == 8) ACM_EPI(8);
    else ACM_EPI(16);
#undef ACM_EPI
    ACM_CHECK_HIP(hipGetLastError());
    if (p->next_x) {                              // the four-rows-per-wave stage does not carry the refill: a launch of its own
        st = acm_dropout(a->n_rows, p->f_in, p->next_x, p->ld_next_x, const_cast<float*>(p->xs), p->ld_xs, p->f_pad, &p->next_drop, stream);
        if (st != ACM_OK) return st;
    }
    return next_projection(a, p, stream);
}

extern "C" int acm_conv_agg_bwd_workspace_bytes(int64_t n_rows, int f_in, int f_out, size_t* bytes) {
    ACM_REQUIRE(bytes, ACM_EINVAL, "acm_conv_agg_bwd_workspace_bytes: NULL argument");
    ACM_REQUIRE(f_in >= 1 && f_in <= 16 && f_out >= 1 && f_out <= 64, ACM_EUNSUPPORTED,
                "acm_conv_agg_bwd_workspace_bytes: f_in %d f_out %d unsupported", f_in, f_out);
    // sized for 4 channels, whole groups of 32, + the following layer's weight gradient (acm_conv_agg_bwd_t.proj_*)
    const size_t npg = ((size_t)3 * f_in * f_out + 12 * (size_t)f_out + 16 + 31) / 32 * 32 + (size_t)6 * f_out;
    *bytes = (size_t)agg_bwd_blocks(n_rows, 3, 0) * npg * sizeof(float);      // the larger of the two grids
    return ACM_OK;
}

extern "C" int acm_conv_agg_bwd(int64_t n_rows, const acm_conv_agg_bwd_t* p, void* workspace,
                                size_t workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(p, ACM_EINVAL, "acm_conv_agg_bwd: NULL argument");
    int st = check_common(p, "acm_conv_agg_bwd");
    if (st != ACM_OK) return st;
    ACM_REQUIRE((p->grad_out || p->proj_dz) && p->agg && p->d_params, ACM_EINVAL, "acm_conv_agg_bwd: NULL tensor pointer");
    ACM_REQUIRE(p->n_channels == 3 || (p->g_struc && p->ld_g_struc >= p->f_out), ACM_EINVAL,
                "acm_conv_agg_bwd: g_struc is NULL / too narrow");
    {
        int64_t ld_max = 64;
        for (int64_t ld : {p->ld_grad_out, p->ld_agg, p->ld_xs, p->ld_head_stats, p->ld_post_scale, p->ld_ps, p->ld_ss, p->ld_g_struc, p->ld_out})
            ld_max = ld > ld_max ? ld : ld_max;
        ACM_REQUIRE(n_rows * ld_max < (int64_t)INT32_MAX, ACM_EUNSUPPORTED, "acm_conv_agg_bwd: too many rows for 32-bit offsets");
    }
    ACM_REQUIRE(((uintptr_t)p->agg) % 16 == 0 && (p->ld_agg * 4) % 16 == 0 && p->ld_agg >= p->f_pad, ACM_EINVAL,
                "acm_conv_agg_bwd: agg rows must be 16-byte aligned and f_pad long");
    size_t need = 0;
    acm_conv_agg_bwd_workspace_bytes(n_rows, p->f_in, p->f_out, &need);
    ACM_REQUIRE(workspace && workspace_bytes >= need, ACM_ENOMEM, "acm_conv_agg_bwd: workspace %zu B < required %zu B",
                workspace_bytes, need);
    const int K = p->n_channels;
    const int npg = 3 * p->f_in * p->f_out + 3 * K * p->f_out + K * K;
    const size_t lds_w = ((size_t)3 * p->f_pad * 64 + 3 * K * 64 + 32 * p->f_pad) * sizeof(float),
                 lds_s = (size_t)4 * npg * sizeof(float);
    const size_t lds = lds_w > lds_s ? lds_w : lds_s;
    ACM_REQUIRE(lds <= 64 * 1024, ACM_EUNSUPPORTED, "acm_conv_agg_bwd: %zu B of LDS needed", lds);
    int nblk = agg_bwd_blocks(n_rows, p->n_channels, lds);
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    if (p->next_agg) {                            // the next step's input gather rides along (header: acm_conv_agg_bwd_t.next_agg)
        const acm_csr_t* na = p->next_a;
        ACM_REQUIRE(na && p->next_xg, ACM_EINVAL, "acm_conv_agg_bwd: next_agg without next_a / next_xg");
        ACM_REQUIRE(p->f_pad == 8 && p->f_out == 64, ACM_EUNSUPPORTED, "acm_conv_agg_bwd: the carried gather needs f_pad 8 and f_out 64");
        const AcmStreams* t = na->streams;
        const int gw = 4;                         // gather waves per workgroup (eight + eight backward waves: 124 us against 110)
        ACM_REQUIRE(t && t->n_waves >= gw && t->n_waves % gw == 0 && t->n_waves <= 256 * gw, ACM_EINVAL,
                    "acm_conv_agg_bwd: next_a needs id streams for a multiple of four waves <= 1024 (acm_csr_build_streams)");
        ACM_REQUIRE(na->vals == nullptr && p->ld_next_xg == 8 && ((uintptr_t)p->next_xg) % 16 == 0 &&
                        na->n_cols * 32 < (int64_t)0xFFFFFFE0u && ((uintptr_t)p->next_agg) % 16 == 0 &&
                        (p->ld_next_agg * 4) % 16 == 0 && p->ld_next_agg >= 8, ACM_EINVAL,
                    "acm_conv_agg_bwd: carried gather: pattern-only operator, 32-byte rows of next_xg, 16-byte aligned next_agg");
        ACM_REQUIRE(p->next_agg != p->agg && p->next_xg != p->xs, ACM_EINVAL, "acm_conv_agg_bwd: next_agg / next_xg alias agg / xs");
        GatherRole gr;
        gr.sv = stream_view(t);
        gr.xg = p->next_xg;
        gr.xg_bytes = (unsigned)(na->n_cols * 32);
        gr.row_scale = p->next_row_scale;
        gr.agg = p->next_agg;
        gr.ld_agg = (long)p->ld_next_agg;
        ACM_REQUIRE(t->n_waves / gw <= agg_bwd_blocks(n_rows, 3, 0), ACM_EUNSUPPORTED,
                    "acm_conv_agg_bwd: %d stream waves for %lld rows (the workspace holds one slab per 16 rows)", t->n_waves, (long long)n_rows);
        nblk = t->n_waves / gw;
        {   // eight-wave workgroups of the sixteen-rows-per-wave backward + the gather role (acm_conv_agg16.hip)
            const int nb16 = acm_agg_bwd16(p, n_rows, partial, agg_bwd_blocks(n_rows, 3, 0), s, &gr, nblk);
            if (nb16 < 0) return -nb16;
            if (nb16 > 0) {
                const int off2 = (npg + 31) / 32 * 32, n2 = p->proj_dz ? 3 * p->f_out * p->proj_f : 0;
                const acm_reduce_seg_t segs[2] = {{partial, nb16, 32, 0, npg, p->d_params, npg, 0, 0, 0, nb16 * 32, 0},
                                                  {partial, nb16, 32, off2, n2, p->proj_d_w, n2 > 0 ? n2 : 1, 0, 0, 0, nb16 * 32, 0}};
                return acm_reduce_emit(p->defer, segs, n2 > 0 ? 2 : 1, s);
            }
        }
        ACM_REQUIRE(false, ACM_EUNSUPPORTED, "acm_conv_agg_bwd: the carried gather (next_agg) needs the sixteen-rows-per-wave backward "
                    "(head_stats, the forward's `out` behind a fused ReLU or no post-op at all, acm_tuning_t.rows16 bit 2)");
    }
    if (!p->next_agg) {                           // sixteen rows per wave, transposed matrix-core layout (acm_conv_agg16.hip)
        const int nb16 = acm_agg_bwd16(p, n_rows, partial, nblk, s, nullptr, 0);
        if (nb16 < 0) return -nb16;
        if (nb16 > 0) {
            const int off2 = (npg + 31) / 32 * 32, n2 = p->proj_dz ? 3 * p->f_out * p->proj_f : 0;
            const acm_reduce_seg_t segs[2] = {{partial, nb16, 32, 0, npg, p->d_params, npg, 0, 0, 0, nb16 * 32, 0},
                                              {partial, nb16, 32, off2, n2, p->proj_d_w, n2 > 0 ? n2 : 1, 0, 0, 0, nb16 * 32, 0}};
            return acm_reduce_emit(p->defer, segs, n2 > 0 ? 2 : 1, s);
        }
    }
    ACM_REQUIRE(!p->proj_dz, ACM_EUNSUPPORTED, "acm_conv_agg_bwd: proj_dz needs the sixteen-rows-per-wave kernel (three channels, "
                "f_pad 8, f_out 64, head_stats, `out`, proj_f <= 2); run acm_proj_bwd and pass grad_out");
#define ACM_BWDK(FPv)                                                                                                  \
    do {                                                                                                              \
        if (K == 3 && p->f_out == 64)                                                                                  \
            hipLaunchKernelGGL((agg_bwd_kernel<FPv, 3, true>), dim3(nblk), dim3(256), lds, s, *p, (int)n_rows, partial); \
        else if (K == 3)                                                                                              \
            hipLaunchKernelGGL((agg_bwd_kernel<FPv, 3, false>), dim3(nblk), dim3(256), lds, s, *p, (int)n_rows, partial); \
        else if (p->f_out == 64)                                                                                      \
            hipLaunchKernelGGL((agg_bwd_kernel<FPv, 4, true>), dim3(nblk), dim3(256), lds, s, *p, (int)n_rows, partial); \
        else                                                                                                          \
            hipLaunchKernelGGL((agg_bwd_kernel<FPv, 4, false>), dim3(nblk), dim3(256), lds, s, *p, (int)n_rows, partial); \
    } while (0)
    if (p->f_pad == 4) ACM_BWDK(4);
    else if (p->f_pad == 8) ACM_BWDK(8);
    else ACM_BWDK(16);
#undef ACM_BWDK
    ACM_CHECK_HIP(hipGetLastError());
    const acm_reduce_seg_t seg = {partial, nblk, 32, 0, npg, p->d_params, npg, 0, 0, 0, nblk * 32, 0};   // d_params[q] = sum_b partial[q / 32][b][q % 32]
    return acm_reduce_emit(p->defer, &seg, 1, s);
}
