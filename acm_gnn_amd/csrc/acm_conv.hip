// Fused ACM layer kernels for gfx950 (MI355X):
//   K2  acm_conv_fwd       one CSR pass over A_low -> all graph channels + adaptive mixing
//   K3  acm_conv_bwd_local row-local backward of the mixing head (+ parameter-gradient reduction)
//   K4  acm_conv_bwd_spmm  transposed SpMM with the high-pass / structure identities folded in
//       acm_spmm           plain CSR x dense (k-hop chains, tests)
//
// Execution shapes (wave = 64 lanes):
//   wide   (F > 8)  one wave per work item, lane l owns columns l, l+64, ... of every channel;
//                   the wave loads 64 (index, value) pairs with one coalesced instruction each,
//                   broadcasts them lane by lane (v_readlane -> SGPR row base) and issues
//                   UNR x NG x NREG independent 256 B row-segment loads before the FMAs.
//   narrow (F <= 8) GS lanes per work item, lanes over *neighbours*, every lane gathers the
//                   whole (NG x F)-float row of its neighbour with vector loads and the group
//                   all-reduces at the end; epilogue runs redundantly in the group.
// Rows longer than `chunk` neighbours are split into several work items whose partial sums
// are combined in slot order by a fix-up kernel (deterministic, no float atomics).
#include "acm_conv_device.h"

int acm_spmm_internal(const acm_csr* a, const void* G, int64_t ldg, int width, float* Y, int64_t ldy,
                      const acm_spmm_opts_t* o, void* workspace, size_t workspace_bytes, acm_stream_t stream,
                      bool* defer_fixup);

// ------------------------------------------------------------------ epilogues
// Layouts A and B give every column exactly one owning lane; layout C replicates the row in
// every lane of the group, so only the group leader stores.
template <class L>
struct Owns {
    static __device__ __forceinline__ bool lane_stores(const L&) { return true; }
};
template <int FP>
struct Owns<LaySerial<FP>> {
    static __device__ __forceinline__ bool lane_stores(const LaySerial<FP>& l) { return l.lead; }
};
template <>
struct Owns<LayPair32> {
    static __device__ __forceinline__ bool lane_stores(const LayPair32& l) { return l.lane < 32; }
};
template <int NB>
struct Owns<LayVec16<NB>> {
    static __device__ __forceinline__ bool lane_stores(const LayVec16<NB>& l) { return l.lane < 16; }
};

struct EpiPlain {
    struct Args {
        float* y;
        long ldy;
        int relu;
        const float* sub;         // optional: y = out_scale[row] * acc - sub_scale[row] * sub[row][col]
        long ld_sub;
        const float* sub_scale;   // optional (NULL = 1)
        const float* out_scale;   // optional (NULL = 1)
    };
    template <class L, int NG>
    static __device__ __forceinline__ void apply(const Args& a, int row, const L& lay, int F,
                                                 const float (&acc)[NG][L::NV]) {
        if (!Owns<L>::lane_stores(lay)) return;
        const float os = a.out_scale ? a.out_scale[row] : 1.f;
#pragma unroll
        for (int i = 0; i < L::NV; ++i) {
            const int col = lay.col(i);
            if (col < F) {
                float v = os * acc[0][i];
                if (a.sub) v -= (a.sub_scale ? a.sub_scale[row] : 1.f) * a.sub[(long)row * a.ld_sub + col];
                a.y[(long)row * a.ldy + col] = a.relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
};

struct EpiFwd {
    using Args = acm_conv_fwd_t;
    template <class L, int NG>
    static __device__ __forceinline__ void apply(const Args& p, int row, const L& lay, int F,
                                                 const float (&acc)[NG][L::NV]) {
        constexpr int NV = L::NV;
        float H[4][NV], hn[4][NV], xhat[4][NV], pre[3][NV];
        const float dg = (NG == 3) ? p.deg[row] : 0.f;
        const float rs = p.row_scale ? p.row_scale[row] : 1.f;      // pattern-only operator: A_low = D^-1 P
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = lay.col(i);
            const bool ok = col < F;
            const float zh = ok ? p.s_high[(long)row * p.ld_s_high + col] : 0.f;
            const float zi = ok ? p.s_mlp[(long)row * p.ld_s_mlp + col] : 0.f;
            const float p0 = rs * acc[0][i];
            const float p1 = zh - rs * acc[1][i];
            pre[0][i] = p0;
            pre[1][i] = p1;
            H[0][i] = p.relu_after ? fmaxf(p0, 0.f) : p0;
            H[1][i] = p.relu_after ? fmaxf(p1, 0.f) : p1;
            H[2][i] = p.relu_mlp ? fmaxf(zi, 0.f) : zi;
            if (NG == 3) {
                const float ss = ok ? p.s_struc[(long)row * p.ld_s_struc + col] : 0.f;
                const float p3 = dg * (rs * acc[NG - 1][i]) - ss;
                pre[2][i] = p3;
                H[3][i] = fmaxf(p3, 0.f);
            } else {
                pre[2][i] = 0.f;
                H[3][i] = 0.f;
            }
            if (!ok) {
                H[0][i] = H[1][i] = H[2][i] = H[3][i] = 0.f;
            }
        }
        HeadOut ho;
        const HeadParams hp = acm_head_params(p);
        acm_head<L, NG + 1>(lay, F, p.layernorm, hp, H, hn, xhat, ho);
        const bool st = Owns<L>::lane_stores(lay);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = lay.col(i);
            if (col < F && st) {
                float o = ho.alpha[0] * H[0][i] + ho.alpha[1] * H[1][i] + ho.alpha[2] * H[2][i];
                if (NG == 3) o += ho.alpha[3] * H[3][i];
                o *= p.scale;
                if (p.post_relu) o = fmaxf(o, 0.f);
                if (p.post_scale) o *= p.post_scale[(long)row * p.ld_post_scale + col];
                if (p.post_drop.p > 0.f) o *= acm_drop1(acm_drop_ctx(p.post_drop), row, col);
                p.out[(long)row * p.ld_out + col] = o;
                float* pr = p.pre + (long)row * p.ld_pre;
                pr[col] = pre[0][i];
                pr[F + col] = pre[1][i];
                if (NG == 3) pr[2 * F + col] = pre[2][i];
            }
        }
        if (lay.leader()) {
            float4 a4 = make_float4(ho.alpha[0], ho.alpha[1], ho.alpha[2], ho.alpha[3]);
            *reinterpret_cast<float4*>(p.att + (long)row * 4) = a4;
        }
    }
};

// Narrow layers (F <= 8), phase 1 of the fused forward: the raw neighbour sums go to the `pre` buffer; phase 2
// (conv_fwd_rows_kernel) finishes every row with one THREAD per row.  In the narrow gather the whole row sits
// in one lane, so running EpiFwd there leaves 31 of 32 lanes idle through the head's exp / rsqrt chains
// (~40 us of the 110 us F = 2 forward on the twitch graph).
struct EpiRaw {
    using Args = acm_conv_fwd_t;
    template <class L, int NG>
    static __device__ __forceinline__ void apply(const Args& p, int row, const L& lay, int F,
                                                 const float (&acc)[NG][L::NV]) {
        if (!Owns<L>::lane_stores(lay)) return;
        float* pr = p.pre + (long)row * p.ld_pre;
#pragma unroll
        for (int i = 0; i < L::NV; ++i) {
            const int col = lay.col(i);
            if (col < F) {
#pragma unroll
                for (int c = 0; c < NG; ++c) pr[c * F + col] = acc[c][i];
            }
        }
    }
};

template <int FP, int NG>
__global__ __launch_bounds__(256) void conv_fwd_rows_kernel(acm_conv_fwd_t p, int n_rows, CsrView csr,
                                                            const float* __restrict__ partial, int row_blocks) {
    const int F = p.f_out;
    float acc[NG][FP];
    if ((int)blockIdx.x >= row_blocks) {
        // tail blocks: the long rows, whose work items left partial sums in the slots -- a 16-lane group per row,
        // lanes over the slots (slot order within a lane, fixed DPP tree across lanes), then the same head.  This is
        // the fix-up pass of the gather folded into this launch (the two parts do not depend on each other).
        const int m = threadIdx.x & 15;
        const int w = ((int)blockIdx.x - row_blocks) * 16 + (threadIdx.x >> 4);
        if (w >= csr.n_long) return;
        const AcmLongRow lr = csr.long_rows[w];
#pragma unroll
        for (int c = 0; c < NG; ++c)
#pragma unroll
            for (int f = 0; f < FP; ++f) acc[c][f] = 0.f;
        for (int s = lr.slot_begin + m; s < lr.slot_end; s += 16) {
            const float* ps = partial + (long)s * (NG * F);
#pragma unroll
            for (int c = 0; c < NG; ++c)
#pragma unroll
                for (int f = 0; f < FP; ++f)
                    if (f < F) acc[c][f] += ps[c * F + f];
        }
#pragma unroll
        for (int c = 0; c < NG; ++c)
#pragma unroll
            for (int f = 0; f < FP; ++f) acc[c][f] = acm_group_sum<16>(acc[c][f]);
        const LaySerial<FP> lay{m == 0};
        EpiFwd::apply<LaySerial<FP>, NG>(p, lr.row, lay, F, acc);
        return;
    }
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n_rows) return;
    if (csr.long_index && csr.long_index[row] >= 0) return;      // done by a tail block
    const float* pr = p.pre + (long)row * p.ld_pre;
#pragma unroll
    for (int c = 0; c < NG; ++c)
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[c][f] = (f < F) ? pr[c * F + f] : 0.f;
    const LaySerial<FP> lay{true};
    EpiFwd::apply<LaySerial<FP>, NG>(p, row, lay, F, acc);     // overwrites this row of `pre` with the final values
}

// The row-local head of a wide three-channel layer as a kernel of its own (acm_conv_head_fwd): the channels' pre-activations
// come from products the caller has already made -- pre_L = g_low[row], pre_H = s_high[row] - g_high[row], Z_I = s_mlp[row]
// (the aggregate-first form for wide inputs: functional._AcmAggWide) -- so there is nothing to gather: a 16-lane group takes a
// row (four rows per wave, 16-byte loads), EpiFwd does the rest exactly as behind a gather.
__global__ __launch_bounds__(256) void conv_head_rows_kernel(acm_conv_fwd_t p, int n_rows) {
    const int lane = threadIdx.x & 63, m = lane & 15;
    const int row = ((int)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    if (row >= n_rows) return;                          // (whole 16-lane groups leave: the group sums stay complete)
    constexpr int F = 64;
    // EpiFwd::apply for NG = 2 with 16-byte loads and stores (the arithmetic, statement for statement, is the same)
    const float4 a = *reinterpret_cast<const float4*>(p.g_low + (long)row * p.ld_g_low + 4 * m);
    const float4 b = *reinterpret_cast<const float4*>(p.g_high + (long)row * p.ld_g_high + 4 * m);
    const float4 zh4 = *reinterpret_cast<const float4*>(p.s_high + (long)row * p.ld_s_high + 4 * m);
    const float4 zi4 = *reinterpret_cast<const float4*>(p.s_mlp + (long)row * p.ld_s_mlp + 4 * m);
    const float aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w};
    const float zh[4] = {zh4.x, zh4.y, zh4.z, zh4.w}, zi[4] = {zi4.x, zi4.y, zi4.z, zi4.w};
    float H[4][4], hn[4][4], xhat[4][4], pre[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float p0 = 1.f * aa[i];
        const float p1 = zh[i] - 1.f * bb[i];
        pre[0][i] = p0, pre[1][i] = p1;
        H[0][i] = p.relu_after ? fmaxf(p0, 0.f) : p0;
        H[1][i] = p.relu_after ? fmaxf(p1, 0.f) : p1;
        H[2][i] = p.relu_mlp ? fmaxf(zi[i], 0.f) : zi[i];
        H[3][i] = 0.f;
    }
    const LayRow16 lay{lane};
    HeadOut ho;
    const HeadParams hp = acm_head_params(p);
    acm_head<LayRow16, 3>(lay, F, p.layernorm, hp, H, hn, xhat, ho);
    float o[4];
    const AcmDropCtx dc = acm_drop_ctx(p.post_drop);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = 4 * m + i;
        float v = ho.alpha[0] * H[0][i] + ho.alpha[1] * H[1][i] + ho.alpha[2] * H[2][i];
        v *= p.scale;
        if (p.post_relu) v = fmaxf(v, 0.f);
        if (p.post_scale) v *= p.post_scale[(long)row * p.ld_post_scale + col];
        if (p.post_drop.p > 0.f) v *= acm_drop1(dc, row, col);
        o[i] = v;
    }
    *reinterpret_cast<float4*>(p.out + (long)row * p.ld_out + 4 * m) = make_float4(o[0], o[1], o[2], o[3]);
    float* pr = p.pre + (long)row * p.ld_pre + 4 * m;
    *reinterpret_cast<float4*>(pr) = make_float4(pre[0][0], pre[0][1], pre[0][2], pre[0][3]);
    *reinterpret_cast<float4*>(pr + F) = make_float4(pre[1][0], pre[1][1], pre[1][2], pre[1][3]);
    if (m == 0) *reinterpret_cast<float4*>(p.att + (long)row * 4) = make_float4(ho.alpha[0], ho.alpha[1], ho.alpha[2], ho.alpha[3]);
}

struct EpiBwd {
    using Args = acm_conv_bwd_spmm_t;
    template <class L, int NG>
    static __device__ __forceinline__ void apply(const Args& p, int row, const L& lay, int F,
                                                 const float (&acc)[NG][L::NV]) {
        if (!Owns<L>::lane_stores(lay)) return;
        const float idg = (NG == 3 && p.inv_deg) ? p.inv_deg[row] : 1.f;
        const float ssc = p.self_scale ? p.self_scale[row] : 1.f;   // pattern-only: s_high holds D^-1 G_H
#pragma unroll
        for (int i = 0; i < L::NV; ++i) {
            const int col = lay.col(i);
            if (col >= F) continue;
            float dl = acc[0][i];
            float dh = ssc * p.s_high[(long)row * p.ld_s_high + col] - acc[1][i];
            if (p.mask_low) dl = (p.mask_low[(long)row * p.ld_mask_low + col] > 0.f) ? dl : 0.f;
            if (p.mask_high) dh = (p.mask_high[(long)row * p.ld_mask_high + col] > 0.f) ? dh : 0.f;
            p.dz_low[(long)row * p.ld_dz_low + col] = dl;
            p.dz_high[(long)row * p.ld_dz_high + col] = dh;
            if (NG == 3)
                p.d_struc[(long)row * p.ld_d_struc + col] =
                    acc[NG - 1][i] - p.s_struc[(long)row * p.ld_s_struc + col] * idg;
        }
    }
};

// The three outputs of K4 do not depend on each other, so the wide backward can run ONE CHANNEL PER PASS: a pass then
// gathers 64-column rows (256 B, two cache lines per neighbour) of a table whose hot part -- the hub rows -- is half as
// large as that of the [G_L | G_H] rows, more of it stays in the 4 MB L2 of an XCD, and the single-channel passes take the
// vector form (scripts/probe_wide.py: 2 x 270 us against 610 us for the 128-column gather on the twitch-shaped graph).
struct EpiBwdLow {
    using Args = acm_conv_bwd_spmm_t;
    template <class L, int NG>
    static __device__ __forceinline__ void apply(const Args& p, int row, const L& lay, int F, const float (&acc)[NG][L::NV]) {
        if (!Owns<L>::lane_stores(lay)) return;
#pragma unroll
        for (int i = 0; i < L::NV; ++i) {
            const int col = lay.col(i);
            if (col >= F) continue;
            float dl = acc[0][i];
            if (p.mask_low) dl = (p.mask_low[(long)row * p.ld_mask_low + col] > 0.f) ? dl : 0.f;
            p.dz_low[(long)row * p.ld_dz_low + col] = dl;
        }
    }
};
struct EpiBwdHigh {
    using Args = acm_conv_bwd_spmm_t;
    template <class L, int NG>
    static __device__ __forceinline__ void apply(const Args& p, int row, const L& lay, int F, const float (&acc)[NG][L::NV]) {
        if (!Owns<L>::lane_stores(lay)) return;
        const float ssc = p.self_scale ? p.self_scale[row] : 1.f;
#pragma unroll
        for (int i = 0; i < L::NV; ++i) {
            const int col = lay.col(i);
            if (col >= F) continue;
            float dh = ssc * p.s_high[(long)row * p.ld_s_high + col] - acc[0][i];
            if (p.mask_high) dh = (p.mask_high[(long)row * p.ld_mask_high + col] > 0.f) ? dh : 0.f;
            p.dz_high[(long)row * p.ld_dz_high + col] = dh;
        }
    }
};
struct EpiBwdStruc {
    using Args = acm_conv_bwd_spmm_t;
    template <class L, int NG>
    static __device__ __forceinline__ void apply(const Args& p, int row, const L& lay, int F, const float (&acc)[NG][L::NV]) {
        if (!Owns<L>::lane_stores(lay)) return;
        const float idg = p.inv_deg ? p.inv_deg[row] : 1.f;
#pragma unroll
        for (int i = 0; i < L::NV; ++i) {
            const int col = lay.col(i);
            if (col < F) p.d_struc[(long)row * p.ld_d_struc + col] = acc[0][i] - p.s_struc[(long)row * p.ld_s_struc + col] * idg;
        }
    }
};

// ------------------------------------------------------------------ wide gather
// element load of the gathered operand: fp32, or bf16 widened to fp32 (exact)
template <bool BF16>
__device__ __forceinline__ float load_gathered(const float* rowp_f32_units, long row_elems, int col) {
    if (BF16) {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(rowp_f32_units) + row_elems + col;
        return __uint_as_float(((unsigned)*p) << 16);
    }
    return rowp_f32_units[row_elems + col];
}

template <int NREG, int NG, int UNR, bool BF16>
__device__ __forceinline__ void gather_wide(const GatherSrc& g, int F, const int32_t* __restrict__ indices,
                                            const float* __restrict__ vals, int begin, int end,
                                            int lane, float (&acc)[NG][NREG]) {
    for (int base = begin; base < end; base += 64) {
        const int kk = base + lane;
        int my_j = 0;
        float my_a = 0.f;
        if (kk < end) {
            my_j = indices[kk];
            my_a = vals ? vals[kk] : 1.f;           // pattern-only operator: implicit ones
        }
        const int cnt = min(64, end - base);  // wave-uniform
        int t = 0;
        for (; t + UNR <= cnt; t += UNR) {
            float z[UNR][NG][NREG];
            float a[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int j = __builtin_amdgcn_readlane(my_j, t + u);
                a[u] = acm_lane_f(my_a, t + u);
#pragma unroll
                for (int c = 0; c < NG; ++c) {
                    const long roff = (long)j * g.ld[c];
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        const int col = lane + 64 * r;
                        z[u][c][r] = (col < F) ? load_gathered<BF16>(g.p[c], roff, col) : 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int c = 0; c < NG; ++c)
#pragma unroll
                    for (int r = 0; r < NREG; ++r) acc[c][r] = fmaf(a[u], z[u][c][r], acc[c][r]);
        }
        for (; t < cnt; ++t) {
            const int j = __builtin_amdgcn_readlane(my_j, t);
            const float a = acm_lane_f(my_a, t);
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                const long roff = (long)j * g.ld[c];
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    const int col = lane + 64 * r;
                    const float z = (col < F) ? load_gathered<BF16>(g.p[c], roff, col) : 0.f;
                    acc[c][r] = fmaf(a, z, acc[c][r]);
                }
            }
        }
    }
}

template <int NREG, int NG, class Epi, bool BF16 = false>
__global__ __launch_bounds__(256) void spmm_wide_kernel(CsrView csr, GatherSrc g, int F,
                                                        typename Epi::Args ea, float* __restrict__ partial) {
    // Blocks take work items in dispatch order (block b -> XCD b % 8): every XCD sees a uniform
    // sample of the rows, and on degree-sorted graphs the heavy items start first.  (A contiguous
    // per-XCD range, the usual GEMM swizzle, left 7 XCDs idle behind the hub rows: 133 -> 327 us.)
    const int lane = threadIdx.x & 63;
    const int w = acm_uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= csr.n_items) return;
    const AcmItem it = csr.items[w];
    const int row = acm_uniform(it.row), begin = acm_uniform(it.begin), end = acm_uniform(it.end),
              slot = acm_uniform(it.slot);
    float acc[NG][NREG];
#pragma unroll
    for (int c = 0; c < NG; ++c)
#pragma unroll
        for (int r = 0; r < NREG; ++r) acc[c][r] = 0.f;
    constexpr int UNR = (NG * NREG >= 8) ? 2 : (NG * NREG >= 4 ? 4 : 8);
    gather_wide<NREG, NG, UNR, BF16>(g, F, csr.indices, csr.vals, begin, end, lane, acc);
    if (slot < 0) {
        LayWide<NREG> lay{lane};
        Epi::template apply<LayWide<NREG>, NG>(ea, row, lay, F, acc);
    } else {
        float* ps = partial + (long)slot * (NG * F);
#pragma unroll
        for (int c = 0; c < NG; ++c)
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const int col = lane + 64 * r;
                if (col < F) ps[c * F + col] = acc[c][r];
            }
    }
}

// bf16 gathered operand, F <= 64 (even): lane l of each half-wave owns the packed column pair (2l, 2l+1), the two
// half-waves walk alternate neighbours, so one dword load per lane fetches 2 neighbours x 128 B per channel --
// half the bytes AND half the load instructions of the fp32 path (2-byte per-lane loads were slower than fp32:
// 804 -> 1290 us).  The halves are combined with v_permlane32_swap, then the epilogue runs in LayPair32.
template <int NG, class Epi, bool BF16>
__global__ __launch_bounds__(256) void spmm_pair_kernel(CsrView csr, GatherSrc g, int F, typename Epi::Args ea,
                                                        float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int w = acm_uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= csr.n_items) return;
    const AcmItem it = csr.items[w];
    const int row = acm_uniform(it.row), begin = acm_uniform(it.begin), end = acm_uniform(it.end),
              slot = acm_uniform(it.slot);
    float acc[NG][2];
#pragma unroll
    for (int c = 0; c < NG; ++c) acc[c][0] = acc[c][1] = 0.f;
    const bool col_ok = 2 * l32 < F;
    constexpr int UNR = 4;
    for (int base = begin; base < end; base += 64) {
        const int kk = base + lane;
        int my_j = 0;
        float my_a = 0.f;
        if (kk < end) {
            my_j = csr.indices[kk];
            my_a = csr.vals ? csr.vals[kk] : 1.f;
        }
        const int cnt = min(64, end - base);
        for (int t = 0; t < cnt; t += 2 * UNR) {
            unsigned zz[UNR][NG];      // bf16: one packed pair
            float2 zf[UNR][NG];        // fp32: the two adjacent columns
            float a[UNR];
            bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int t0 = t + 2 * u;                         // wave-uniform
                const bool have0 = t0 < cnt, have1 = t0 + 1 < cnt;
                const int j0 = __builtin_amdgcn_readlane(my_j, have0 ? t0 : 0);
                const int j1 = __builtin_amdgcn_readlane(my_j, have1 ? t0 + 1 : 0);
                const float a0 = acm_lane_f(my_a, have0 ? t0 : 0), a1 = acm_lane_f(my_a, have1 ? t0 + 1 : 0);
                const int j = half ? j1 : j0;
                a[u] = half ? a1 : a0;
                ok[u] = (half ? have1 : have0) && col_ok;
#pragma unroll
                for (int c = 0; c < NG; ++c) {
                    if (BF16) {
                        const unsigned* rowp = reinterpret_cast<const unsigned*>(
                            reinterpret_cast<const unsigned short*>(g.p[c]) + (long)j * g.ld[c]);
                        zz[u][c] = rowp[col_ok ? l32 : 0];
                    } else {
                        const float2* rowp = reinterpret_cast<const float2*>(g.p[c] + (long)j * g.ld[c]);
                        zf[u][c] = rowp[col_ok ? l32 : 0];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int c = 0; c < NG; ++c) {
                    const float lo = BF16 ? __uint_as_float(zz[u][c] << 16) : zf[u][c].x;
                    const float hi = BF16 ? __uint_as_float(zz[u][c] & 0xFFFF0000u) : zf[u][c].y;
                    acc[c][0] = ok[u] ? fmaf(a[u], lo, acc[c][0]) : acc[c][0];
                    acc[c][1] = ok[u] ? fmaf(a[u], hi, acc[c][1]) : acc[c][1];
                }
        }
    }
    // add the two half-waves (fixed order: lower + upper)
#pragma unroll
    for (int c = 0; c < NG; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const acm_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[c][i]), __float_as_uint(acc[c][i]),
                                                                 false, false);
            acc[c][i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
    if (slot < 0) {
        LayPair32 lay{lane};
        Epi::template apply<LayPair32, NG>(ea, row, lay, F, acc);
    } else if (lane < 32) {
        float* ps = partial + (long)slot * (NG * F);
#pragma unroll
        for (int c = 0; c < NG; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int col = 2 * l32 + i;
                if (col < F) ps[c * F + col] = acc[c][i];
            }
    }
}

// ------------------------------------------------------------------ wide gather, vector form
// Four neighbours per load instruction: a 16-lane group (one DPP row) fetches one 64-column block of one neighbour's
// row with a dwordx4 per lane (256 B per group, 1 KB per wave instruction -- the dword-per-lane form above moves 256 B
// per instruction), the four groups of the wave walk four consecutive neighbours.  Column ids are loaded TRANSPOSED
// (lane (q, m) holds neighbour 4 m + q of the 64-id batch) so that step u needs lane u of every group: one
// `v_mov_b32_dpp row_newbcast:u` per step, no readlane / select chain and no LDS crossbar.  UNR steps x NG channels x NB
// blocks of loads are in flight per wave (8 KB at NG = 2), the four groups' partial sums meet at the end through
// v_permlane16/32_swap (fixed order), and the epilogue runs in LayVec16 (lane m owns columns 4 m .. 4 m + 3 of every
// block).  Needs 16-byte aligned rows (F % 4 == 0, ld % 4 == 0); row offsets are 32-bit byte offsets (table < 4 GB).
// B16: the gathered tables hold bf16 (acm_cast_bf16): the lane's four columns are one 8-byte fetch (a 64-column row is ONE
// 128-byte line instead of two), widened exactly to fp32 -- same lane layout, same fp32 sums.
template <int NG, int NB, int UNR, int BLK, bool B16 = false>
__device__ __forceinline__ void gather_vec_block(const GatherSrc& g, const unsigned (&ldb)[3], const unsigned (&blk_off)[NB],
                                                 unsigned ok_mask, int my_j, float my_a, float (&acc)[NG][4 * NB]) {
    float4 z[UNR][NG][NB];
    float a[UNR];
#pragma unroll
    for (int uu = 0; uu < UNR; ++uu) {
        const unsigned j = (unsigned)acm_row_bcast(my_j, BLK * UNR + uu);
        a[uu] = __int_as_float(acm_row_bcast(__float_as_int(my_a), BLK * UNR + uu));
#pragma unroll
        for (int c = 0; c < NG; ++c) {
            const char* rp = reinterpret_cast<const char*>(g.p[c]) + (size_t)(j * ldb[c]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (B16) {
                    const uint2 w = *reinterpret_cast<const uint2*>(rp + blk_off[b]);
                    z[uu][c][b] = make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u),
                                              __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xFFFF0000u));
                } else {
                    z[uu][c][b] = *reinterpret_cast<const float4*>(rp + blk_off[b]);
                }
            }
        }
    }
#pragma unroll
    for (int uu = 0; uu < UNR; ++uu) {
        // an idle slot (beyond the row's end) fetched row 0 and a lane whose columns lie beyond F fetched the row's first
        // bytes: select, never multiply by a zero weight (0 * inf = NaN would leak a non-finite row the operator does not
        // reference)
        const float av = a[uu];
#pragma unroll
        for (int c = 0; c < NG; ++c)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const bool live = ((ok_mask >> b) & 1u) && av != 0.f;
                acc[c][4 * b + 0] = live ? fmaf(av, z[uu][c][b].x, acc[c][4 * b + 0]) : acc[c][4 * b + 0];
                acc[c][4 * b + 1] = live ? fmaf(av, z[uu][c][b].y, acc[c][4 * b + 1]) : acc[c][4 * b + 1];
                acc[c][4 * b + 2] = live ? fmaf(av, z[uu][c][b].z, acc[c][4 * b + 2]) : acc[c][4 * b + 2];
                acc[c][4 * b + 3] = live ? fmaf(av, z[uu][c][b].w, acc[c][4 * b + 3]) : acc[c][4 * b + 3];
            }
    }
}

template <int NG, int NB, class Epi, bool B16 = false>
__global__ __launch_bounds__(256) void spmm_vec_kernel(CsrView csr, GatherSrc g, int F, typename Epi::Args ea,
                                                       float* __restrict__ partial) {
    constexpr int UNR = (NG * NB >= 4) ? 2 : 4;          // 8 (NG * NB <= 2), 12 (NG = 3) or NG * NB * 2 loads in flight
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    const int w = acm_uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= csr.n_items) return;
    const AcmItem it = csr.items[w];
    const int row = acm_uniform(it.row), begin = acm_uniform(it.begin), end = acm_uniform(it.end),
              slot = acm_uniform(it.slot);
    float acc[NG][4 * NB];
#pragma unroll
    for (int c = 0; c < NG; ++c)
#pragma unroll
        for (int i = 0; i < 4 * NB; ++i) acc[c][i] = 0.f;
    // byte offset of the lane's float4 in column block b; a lane whose columns lie beyond F (F % 4 == 0, F < 64 NB) reads
    // the row's first bytes instead (always inside the row) and contributes nothing
    unsigned blk_off[NB], ok_mask = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const bool ok = 64 * b + 4 * m < F;
        blk_off[b] = ok ? (B16 ? 128u * b + 8u * m : 256u * b + 16u * m) : 0u;
        ok_mask |= ok ? (1u << b) : 0u;
    }
    unsigned ldb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) ldb[c] = c < NG ? (unsigned)g.ld[c] * (B16 ? 2u : 4u) : 0u;
    const int pos = 4 * m + q;                           // transposed id layout (see above)
    for (int base = begin; base < end; base += 64) {
        const int cnt = min(64, end - base);             // wave-uniform
        int my_j = 0;
        float my_a = 0.f;
        if (pos < cnt) {
            my_j = csr.indices[base + pos];
            my_a = csr.vals ? csr.vals[base + pos] : 1.f;
        }
        const int steps = (cnt + 3) >> 2;
        // the step index must be a compile-time constant for the DPP broadcast: 16 / UNR unrolled blocks, uniform exits
#define ACM_VEC_BLK(B)                                                                                          \
        if (B * UNR < steps) gather_vec_block<NG, NB, UNR, B, B16>(g, ldb, blk_off, ok_mask, my_j, my_a, acc)
        ACM_VEC_BLK(0);
        ACM_VEC_BLK(1);
        ACM_VEC_BLK(2);
        ACM_VEC_BLK(3);
        if (UNR == 2) {
            ACM_VEC_BLK(4);
            ACM_VEC_BLK(5);
            ACM_VEC_BLK(6);
            ACM_VEC_BLK(7);
        }
#undef ACM_VEC_BLK
    }
#pragma unroll
    for (int c = 0; c < NG; ++c)
#pragma unroll
        for (int i = 0; i < 4 * NB; ++i) acc[c][i] = acm_cross_row_sum(acc[c][i]);
    if (slot < 0) {
        LayVec16<NB> lay{lane};
        Epi::template apply<LayVec16<NB>, NG>(ea, row, lay, F, acc);
    } else if (q == 0) {
        float* ps = partial + (long)slot * (NG * F);
#pragma unroll
        for (int c = 0; c < NG; ++c)
#pragma unroll
            for (int i = 0; i < 4 * NB; ++i) {
                const int col = 64 * (i >> 2) + 4 * m + (i & 3);
                if (col < F) ps[c * F + col] = acc[c][i];
            }
    }
}

// One wave per long row: add its partial slots in slot order, then the epilogue.
template <int NREG, int NG, class Epi>
__global__ __launch_bounds__(256) void spmm_fixup_kernel(CsrView csr, int F, typename Epi::Args ea,
                                                         const float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int w = acm_uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= csr.n_long) return;
    const AcmLongRow lr = csr.long_rows[w];
    const int row = acm_uniform(lr.row), sb = acm_uniform(lr.slot_begin), se = acm_uniform(lr.slot_end);
    float acc[NG][NREG];
#pragma unroll
    for (int c = 0; c < NG; ++c)
#pragma unroll
        for (int r = 0; r < NREG; ++r) acc[c][r] = 0.f;
    // four slots' loads in flight, added in slot order (a dependent load per slot made this the latency of 16 round trips)
    int s = sb;
    for (; s + 4 <= se; s += 4) {
        float v[4][NG][NREG];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* ps = partial + (long)(s + u) * (NG * F);
#pragma unroll
            for (int c = 0; c < NG; ++c)
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    const int col = lane + 64 * r;
                    v[u][c][r] = col < F ? ps[c * F + col] : 0.f;
                }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < NG; ++c)
#pragma unroll
                for (int r = 0; r < NREG; ++r) acc[c][r] += v[u][c][r];
    }
    for (; s < se; ++s) {
        const float* ps = partial + (long)s * (NG * F);
#pragma unroll
        for (int c = 0; c < NG; ++c)
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const int col = lane + 64 * r;
                if (col < F) acc[c][r] += ps[c * F + col];
            }
    }
    LayWide<NREG> lay{lane};
    Epi::template apply<LayWide<NREG>, NG>(ea, row, lay, F, acc);
}

// ------------------------------------------------------------------ narrow gather
template <int FP>
__device__ __forceinline__ void load_row(const float* __restrict__ p, int F, bool vec, float (&z)[FP]) {
    if (vec) {
        if (FP == 2) {
            const float2 v = *reinterpret_cast<const float2*>(p);
            z[0] = v.x;
            z[1] = v.y;
        } else {
#pragma unroll
            for (int q = 0; q < FP / 4; ++q) {
                const float4 v = reinterpret_cast<const float4*>(p)[q];
                z[4 * q + 0] = v.x;
                z[4 * q + 1] = v.y;
                z[4 * q + 2] = v.z;
                z[4 * q + 3] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int f = 0; f < FP; ++f) z[f] = (f < F) ? p[f] : 0.f;
    }
}

// Narrow gather (F <= 8): one neighbour per lane, GS lanes per work item, the whole gathered row in the lane.
// MERGED: channels 0 and 1 are one contiguous 16-byte-aligned block [c0 (FP) | c1 (FP)] in a row of
// g.p[0], fetched with float4 loads -- one L2 request per neighbour instead of two (the narrow
// kernels are bound by L1->L2 request count, profiles/r01_pmc_*.csv).
// Software-pipelined over the work list: most rows of a power-law graph are one step long (79 % of the
// twitch rows have <= 64 neighbours), so the per-item chain
//     item descriptor -> column ids -> gathered rows -> reduce -> epilogue
// is four dependent memory latencies with nothing to overlap them inside the wave.  A group therefore walks
// the work list with a grid stride (grid capped at NARROW_MAX_BLOCKS), and while the rows of the current
// step are in flight it already has the next item's descriptor and the next step's column ids / values
// requested (of the same item, or of the next one when this was its last step).
constexpr int NARROW_MAX_BLOCKS = 8192;
#ifndef ACM_NARROW_U
#define ACM_NARROW_U 2
#endif

// U = neighbours per lane and step (rows in flight per lane): a lane takes neighbours gl, gl + GS, gl + 2 GS, ... of its item in
// that order whatever U is, so U changes how many steps an item takes -- the dependent chain of a long item -- and not one bit
// of the result.  Measured (round 4, profiles/r04_narrow_u4.txt): U = 4 changes neither the single-GPU kernels (70.4 / 56.0 us
// against 72 / 54.6) nor a rank's kernels of the 8-rank plan (33.0 us against 32.9): the sixteen pieces of the longest row
// are bound by the texture path of the ONE CU their window runs on, not by the number of dependent steps.
template <int FP, int NG, int GS, bool MERGED, class Epi, int U = ACM_NARROW_U>
__global__ __launch_bounds__(256) void spmm_narrow_kernel(CsrView csr, GatherSrc g, int F, int vecmask,
                                                               typename Epi::Args ea, float* __restrict__ partial) {
    constexpr int GPB = 256 / GS;
    // GS == 16: a workgroup round is one window of the work list, so the pieces of a long row meet in LDS and the
    // first piece's group finishes the row -- no partial slots, no fix-up launch (acm_csr.cpp, build_items)
    constexpr bool COOP = GS == ACM_WINDOW && GPB == ACM_WINDOW;
    __shared__ float coop_lds[COOP ? ACM_WINDOW * NG * FP : 1];
    const int gl = threadIdx.x % GS;
    const int G = gridDim.x * GPB;
    int w = blockIdx.x * GPB + threadIdx.x / GS;
    if (w >= csr.n_items) return;
    AcmItem it = csr.items[w];
    int k0 = it.begin;
    const bool unit = csr.vals == nullptr;
    bool v[U];
    int j[U];
    float a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = k0 + gl + u * GS;
        v[u] = k < it.end;
        j[u] = v[u] ? csr.indices[k] : 0;
        a[u] = v[u] ? (unit ? 1.f : csr.vals[k]) : 0.f;
    }
    while (true) {
        const int wn = w + G;
        const bool has_next = wn < csr.n_items;
        AcmItem itn = it;
        if (has_next) itn = csr.items[wn];
        float acc[NG][FP];
#pragma unroll
        for (int c = 0; c < NG; ++c)
#pragma unroll
            for (int f = 0; f < FP; ++f) acc[c][f] = 0.f;
        while (true) {
            float z[U][NG][FP];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MERGED) {
                    float t[2 * FP];
                    load_row<2 * FP>(g.p[0] + (long)j[u] * g.ld[0], 2 * FP, true, t);
#pragma unroll
                    for (int f = 0; f < FP; ++f) {
                        z[u][0][f] = t[f];
                        if (NG > 1) z[u][1 % NG][f] = t[FP + f];
                    }
#pragma unroll
                    for (int c = 2; c < NG; ++c) load_row<FP>(g.p[c] + (long)j[u] * g.ld[c], F, (vecmask >> c) & 1, z[u][c]);
                } else {
#pragma unroll
                    for (int c = 0; c < NG; ++c) load_row<FP>(g.p[c] + (long)j[u] * g.ld[c], F, (vecmask >> c) & 1, z[u][c]);
                }
            }
            // requests of the next step, issued before the rows above are consumed
            const int k1 = k0 + U * GS;
            const bool more = k1 < it.end;
            const int pb = more ? k1 : itn.begin;
            const int pe = more ? it.end : (has_next ? itn.end : pb);
            bool nv[U];
            int nj[U];
            float na[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = pb + gl + u * GS;
                nv[u] = k < pe;
                nj[u] = nv[u] ? csr.indices[k] : 0;
                na[u] = nv[u] ? (unit ? 1.f : csr.vals[k]) : 0.f;
            }
#pragma unroll
            for (int c = 0; c < NG; ++c)
#pragma unroll
                for (int f = 0; f < FP; ++f)
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[c][f] = v[u] ? fmaf(a[u], z[u][c][f], acc[c][f]) : acc[c][f];
#pragma unroll
            for (int u = 0; u < U; ++u) j[u] = nj[u], a[u] = na[u], v[u] = nv[u];
            if (!more) break;
            k0 = k1;
        }
#pragma unroll
        for (int c = 0; c < NG; ++c)
#pragma unroll
            for (int f = 0; f < FP; ++f) acc[c][f] = acm_group_sum<GS>(acc[c][f]);
        if (COOP && w / ACM_WINDOW < csr.n_windows) {          // uniform over the workgroup: every item here is a piece
            const int g = threadIdx.x / GS;
            if (gl == 0) {
#pragma unroll
                for (int c = 0; c < NG; ++c)
#pragma unroll
                    for (int f = 0; f < FP; ++f) coop_lds[(g * NG + c) * FP + f] = acc[c][f];
            }
            __syncthreads();
            const AcmLongRow lr = csr.long_rows[csr.long_index[it.row]];
            // a row of several windows (acm_csr.cpp, build_items) fills this window alone: its sum goes to the slot of the
            // window's first piece and spmm_fixup_windows_kernel adds the windows
            const bool multi = lr.windows > 1;
            if (multi ? g == 0 : it.slot == lr.slot_begin) {    // first piece: add the others in slot order
                const int pieces = multi ? ACM_WINDOW : lr.slot_end - lr.slot_begin;
#pragma unroll
                for (int c = 0; c < NG; ++c)
#pragma unroll
                    for (int f = 0; f < FP; ++f) {
                        float t = 0.f;
                        for (int q = 0; q < pieces; ++q) t += coop_lds[((g + q) * NG + c) * FP + f];
                        acc[c][f] = t;
                    }
                if (!multi) {
                    LaySerial<FP> lay{gl == 0};
                    Epi::template apply<LaySerial<FP>, NG>(ea, it.row, lay, F, acc);
                } else if (gl == 0) {
                    float* ps = partial + (long)it.slot * (NG * F);
#pragma unroll
                    for (int c = 0; c < NG; ++c)
#pragma unroll
                        for (int f = 0; f < FP; ++f)
                            if (f < F) ps[c * F + f] = acc[c][f];
                }
            }
            __syncthreads();
        } else if (it.slot < 0) {
            LaySerial<FP> lay{gl == 0};
            Epi::template apply<LaySerial<FP>, NG>(ea, it.row, lay, F, acc);
        } else if (gl == 0) {
            float* ps = partial + (long)it.slot * (NG * F);
#pragma unroll
            for (int c = 0; c < NG; ++c)
#pragma unroll
                for (int f = 0; f < FP; ++f)
                    if (f < F) ps[c * F + f] = acc[c][f];
        }
        if (!has_next) break;
        it = itn;
        w = wn;
        k0 = it.begin;
    }
}

// Long rows of the narrow path: a 16-lane group per row, lanes over the partial slots (the wide
// fix-up would leave 62 of 64 lanes idle at F = 2 and chain up to deg/chunk dependent loads).
template <int FP, int NG, class Epi>
__global__ __launch_bounds__(256) void spmm_fixup_narrow_kernel(CsrView csr, int F, typename Epi::Args ea,
                                                                const float* __restrict__ partial) {
    const int m = threadIdx.x & 15;
    const int w = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (w >= csr.n_long) return;
    const AcmLongRow lr = csr.long_rows[w];
    float acc[NG][FP];
#pragma unroll
    for (int c = 0; c < NG; ++c)
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[c][f] = 0.f;
    for (int s = lr.slot_begin + m; s < lr.slot_end; s += 16) {
        const float* ps = partial + (long)s * (NG * F);
#pragma unroll
        for (int c = 0; c < NG; ++c)
#pragma unroll
            for (int f = 0; f < FP; ++f)
                if (f < F) acc[c][f] += ps[c * F + f];
    }
#pragma unroll
    for (int c = 0; c < NG; ++c)
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[c][f] = acm_group_sum<16>(acc[c][f]);
    LaySerial<FP> lay{m == 0};
    Epi::template apply<LaySerial<FP>, NG>(ea, lr.row, lay, F, acc);
}

// Rows of several windows (AcmLongRow.windows > 1) after a narrow gather with sixteen lanes per item: window q of the row
// left its sum in slot slot_begin + 16 q; lane q of a 16-lane group fetches it, one group sum (fixed order), epilogue.
template <int FP, int NG, class Epi>
__global__ __launch_bounds__(256) void spmm_fixup_windows_kernel(CsrView csr, int F, typename Epi::Args ea,
                                                                 const float* __restrict__ partial) {
    const int m = threadIdx.x & 15;
    const int w = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (w >= csr.n_long) return;
    const AcmLongRow lr = csr.long_rows[w];
    if (lr.windows <= 1) return;
    float acc[NG][FP];
    const float* ps = partial + (long)(lr.slot_begin + ACM_WINDOW * m) * (NG * F);
#pragma unroll
    for (int c = 0; c < NG; ++c)
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[c][f] = (m < lr.windows && f < F) ? ps[c * F + f] : 0.f;
#pragma unroll
    for (int c = 0; c < NG; ++c)
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[c][f] = acm_group_sum<16>(acc[c][f]);
    LaySerial<FP> lay{m == 0};
    Epi::template apply<LaySerial<FP>, NG>(ea, lr.row, lay, F, acc);
}

// The four-channel narrow gather (structure_info = 1, F = 2: the output layer of the reference's two-class models) over
// PACKED 32-byte rows [c0 c0 c1 c1 | c2 c2 - -]: with the third gathered channel in a table of its own a neighbour costs two
// fetches to two distinct lines (the [c0 | c1] block and the 8-byte c2 row) -- 115 us on the twitch-shaped graph against
// 51 us for the three-channel layer -- while what bounds these gathers is the number of distinct lines per wave
// instruction, not their width (ta_rate).  Two adjacent lanes fetch the two 16-byte halves of a
// neighbour's row (the form of agg_fused_pair_kernel): one line per neighbour again.
// Lane (e = gl >> 1, h = gl & 1) of the 16-lane group: neighbours k0 + e + 8 u (u = 0..3), half h of the row.
template <class Epi>
__global__ __launch_bounds__(256) void spmm_narrow_pair3_kernel(CsrView csr, const float* __restrict__ table, typename Epi::Args ea,
                                                                float* __restrict__ partial) {
    constexpr int FP = 2, NG = 3, GPB = 16, U = 4, STEP = 8 * U;
    static_assert(GPB == ACM_WINDOW, "one window of work items per workgroup round");
    __shared__ float coop[ACM_WINDOW * 8];
    const int gl = threadIdx.x & 15, e = gl >> 1, h = gl & 1;
    const int G = gridDim.x * GPB;
    int w = blockIdx.x * GPB + (threadIdx.x >> 4);
    if (w >= csr.n_items) return;
    const bool unit = csr.vals == nullptr;
    const float* th = table + 4 * h;
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
            for (int u = 0; u < U; ++u) z[u] = *reinterpret_cast<const float4*>(th + (long)j[u] * 8);
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
        if (w / ACM_WINDOW < csr.n_windows) {                   // a window of pieces: they meet in LDS (see spmm_narrow_kernel)
            const int g = threadIdx.x >> 4;
            if (gl < 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) coop[g * 8 + 4 * h + i] = acc[i];
            }
            __syncthreads();
            const AcmLongRow lr = csr.long_rows[csr.long_index[it.row]];
            const bool multi = lr.windows > 1;                  // a row of several windows: see spmm_narrow_kernel
            finish = multi ? g == 0 : it.slot == lr.slot_begin;
            if (finish && gl < 2) {
                const int pieces = multi ? ACM_WINDOW : lr.slot_end - lr.slot_begin;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = 0.f;
                    for (int q = 0; q < pieces; ++q) t += coop[(g + q) * 8 + 4 * h + i];
                    acc[i] = t;
                }
                if (multi) {                                    // [c0 c0 c1 c1] from half 0, [c2 c2] from half 1: slot layout c * 2 + f
                    float* ps = partial + (long)it.slot * (NG * FP) + 4 * h;
                    ps[0] = acc[0], ps[1] = acc[1];
                    if (h == 0) ps[2] = acc[2], ps[3] = acc[3];
                }
            }
            if (multi) finish = false;
            __syncthreads();
        }
        // lane 0 of the group: [c0 | c1] are its own sums, c2 its neighbour's (the other half of the row)
        const float s0 = acm_dpp<0xB1>(acc[0]), s1 = acm_dpp<0xB1>(acc[1]);      // quad_perm [1,0,3,2]
        if (finish) {
            float out[NG][FP] = {{acc[0], acc[1]}, {acc[2], acc[3]}, {s0, s1}};
            LaySerial<FP> lay{gl == 0};
            Epi::template apply<LaySerial<FP>, NG>(ea, it.row, lay, 2, out);
        }
        if (!has_next) break;
        it = itn;
        w = wn;
        k0 = it.begin;
    }
}

// ------------------------------------------------------------------ host-side dispatch
// acm_conv_local16.hip: K3 at F = 64, k = 3 with sixteen rows per wave
int acm_bwd_local16(const acm_conv_bwd_local_t* p, int64_t n_rows, float* partial, int max_blocks, hipStream_t s);

namespace {

// lanes per work item of the narrow gather: 8 for very sparse graphs, 16 up to an average degree of 160 (a power-law
// graph with mean 82 has median 30: with 32 lanes x 2 neighbours most lanes of most rows idle), 32 beyond
int narrow_lanes(const acm_csr* a) {
    const double avg = (double)a->nnz / (double)(a->n_rows > 0 ? a->n_rows : 1);
    return avg <= 12.0 ? 8 : (avg <= 160.0 ? 16 : 32);
}
// with 16 lanes per item a workgroup round is one window: the narrow gather finishes the long rows itself
bool narrow_finishes_long_rows(const acm_csr* a) { return narrow_lanes(a) == ACM_WINDOW; }

// (spmm_narrow_pair3_kernel exists for three gathered channels only; other NG never reach the call)
template <int NG, class Epi>
void launch_pair3(int grid, hipStream_t st, const CsrView& v, const float* table, const typename Epi::Args& ea, float* partial) {
    if constexpr (NG == 3) hipLaunchKernelGGL((spmm_narrow_pair3_kernel<Epi>), dim3(grid), dim3(256), 0, st, v, table, ea, partial);
}

// after a narrow gather with sixteen lanes per item: the rows of several windows (none on most operators)
template <int NG, class Epi>
int finish_window_rows(const acm_csr* a, const CsrView& v, int F, const typename Epi::Args& ea, const float* partial, hipStream_t st) {
    if (a->n_multi == 0) return ACM_OK;
    const int grid = (int)((a->n_long + 15) / 16);
    if (F <= 2) hipLaunchKernelGGL((spmm_fixup_windows_kernel<2, NG, Epi>), dim3(grid), dim3(256), 0, st, v, F, ea, partial);
    else if (F <= 4) hipLaunchKernelGGL((spmm_fixup_windows_kernel<4, NG, Epi>), dim3(grid), dim3(256), 0, st, v, F, ea, partial);
    else hipLaunchKernelGGL((spmm_fixup_windows_kernel<8, NG, Epi>), dim3(grid), dim3(256), 0, st, v, F, ea, partial);
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}

template <int NG, class Epi>
int launch_gather(const acm_csr* a, const GatherSrc& g, int F, const typename Epi::Args& ea,
                  void* workspace, size_t ws_bytes, hipStream_t st, const char* who,
                  const float* vals_override = nullptr, bool bf16 = false, bool defer_fixup = false) {
    const size_t need = (size_t)a->n_slots * (size_t)(NG * F) * sizeof(float);
    ACM_REQUIRE(ws_bytes >= need && (need == 0 || workspace), ACM_ENOMEM,
                "%s: workspace %zu B < required %zu B", who, ws_bytes, need);
    float* partial = (float*)workspace;
    CsrView v = acm_view(a);
    if (vals_override) v.vals = vals_override;
    if (a->n_items == 0) return ACM_OK;
    ACM_REQUIRE(!bf16 || (F > 8 && F <= 64 && F % 2 == 0), ACM_EUNSUPPORTED,
                "%s: bf16 gathered operands are implemented for even 8 < F <= 64", who);
    if (bf16) {
        bool aligned = true;
        for (int c = 0; c < NG; ++c) aligned = aligned && ((uintptr_t)g.p[c]) % 4 == 0 && g.ld[c] % 2 == 0;
        ACM_REQUIRE(aligned, ACM_EINVAL, "%s: bf16 operands must be 4-byte aligned with an even leading dimension", who);
    }
    if (F <= 8) {
        int vecmask = 0;
        const int FP = F <= 2 ? 2 : (F <= 4 ? 4 : 8);
        for (int c = 0; c < NG; ++c) {
            const size_t al = (FP == 2) ? 8 : 16;
            // F < FP: a row pitch of at least FP columns lets the fetch read the whole block (what lies beyond F lands in
            // accumulator columns no epilogue looks at); the operand must cover n_cols x ld floats (acm_hip.h)
            const bool ok = (F == FP || g.ld[c] >= FP) && (((uintptr_t)g.p[c]) % al == 0) &&
                            ((g.ld[c] * sizeof(float)) % al == 0);
            vecmask |= ok ? (1 << c) : 0;
        }
        const int gs = narrow_lanes(a);
        // three gathered channels of two columns each in packed 32-byte rows [c0 c0 c1 c1 | c2 c2 - -]: the pair-lane kernel
        if (NG == 3 && F == 2 && gs == 16 && !bf16 && g.p[1] == g.p[0] + 2 && g.p[2] == g.p[0] + 4 && g.ld[0] == 8 && g.ld[1] == 8 &&
            g.ld[2] == 8 && ((uintptr_t)g.p[0]) % 32 == 0 && (a->n_long == 0 || a->long_index != nullptr)) {
            int grid = (int)((a->n_items + 15) / 16);
            if (grid > NARROW_MAX_BLOCKS) grid = NARROW_MAX_BLOCKS;
            launch_pair3<NG, Epi>(grid, st, v, g.p[0], ea, partial);
            ACM_CHECK_HIP(hipGetLastError());
            return finish_window_rows<NG, Epi>(a, v, F, ea, partial, st);
        }
        // [channel 0 | channel 1] contiguous and block-aligned => one vector fetch for both
        // (F < FP: the channels are blocks of FP columns, [c0 pad | c1 pad]; what the fetch reads beyond F lands in
        // accumulator columns no epilogue looks at)
        const bool merged = NG >= 2 && g.p[1] == g.p[0] + FP && g.ld[0] == g.ld[1] &&
                            ((uintptr_t)g.p[0]) % (8 * FP) == 0 && (g.ld[0] * sizeof(float)) % (8 * FP) == 0;
#define ACM_NARROW(FPv, GSv)                                                                            \
    do {                                                                                                \
        const int gpb = 256 / GSv;                                                                      \
        int grid = (int)((a->n_items + gpb - 1) / gpb);                                                 \
        if (grid > NARROW_MAX_BLOCKS) grid = NARROW_MAX_BLOCKS;                                         \
        if (merged)                                                                                     \
            hipLaunchKernelGGL((spmm_narrow_kernel<FPv, NG, GSv, (NG >= 2), Epi>), dim3(grid), dim3(256), 0, \
                               st, v, g, F, vecmask, ea, partial);                                      \
        else                                                                                            \
            hipLaunchKernelGGL((spmm_narrow_kernel<FPv, NG, GSv, false, Epi>), dim3(grid), dim3(256), 0, \
                               st, v, g, F, vecmask, ea, partial);                                      \
    } while (0)
        if (FP == 2) {
            if (gs == 8) ACM_NARROW(2, 8); else if (gs == 16) ACM_NARROW(2, 16); else ACM_NARROW(2, 32);
        } else if (FP == 4) {
            if (gs == 8) ACM_NARROW(4, 8); else if (gs == 16) ACM_NARROW(4, 16); else ACM_NARROW(4, 32);
        } else {
            if (gs == 8) ACM_NARROW(8, 8); else if (gs == 16) ACM_NARROW(8, 16); else ACM_NARROW(8, 32);
        }
#undef ACM_NARROW
    } else {
        ACM_REQUIRE(F <= 256, ACM_EUNSUPPORTED, "%s: F = %d > 256 columns per channel", who, F);
        const int grid = (int)((a->n_items + 3) / 4);
        // fp32 rows of 34..64 columns whose gathered matrices fit the L2 (Squirrel / Chameleon / Cora sizes): 32 lanes x
        // float2 cover a row, so the two half-waves take two neighbours per instruction -- the wide kernel spends one
        // load + one FMA instruction per neighbour on a half-empty wave and is issue-bound there (81 -> 69 us on
        // Squirrel).  On the 168k-node graph the same gather is bound by the Infinity-Cache fills and the pair form is
        // 5-10 % slower, so it is not used.
        bool pair32 = !bf16 && F > 32 && F <= 64 && F % 2 == 0 && (size_t)a->n_cols * F * NG * sizeof(float) <= (8u << 20);
        for (int c = 0; c < NG && pair32; ++c) pair32 = ((uintptr_t)g.p[c]) % 8 == 0 && g.ld[c] % 2 == 0;
        // vector form: 16-byte aligned rows, 32-bit byte offsets into the gathered tables
        // Measured on the twitch-shaped graph (scripts/probe_wide.py, profiles/r02_probe_wide.txt): rows served by the L2
        // come at 21 TB/s through the vector form against 12 TB/s, rows from the Infinity Cache at 7.5 TB/s through
        // either -- the fabric, not the load instruction, bounds the large-graph gathers.  With a fused head (EpiFwd) the
        // vector layout runs the head four times redundantly, and with two gathered channels its 58 VGPRs cost
        // occupancy, so it is the default for single-channel products (k-hop chains, spmm_sub, the S gather of the
        // aggregate-first structure channel); acm_tuning_t.wide_form = 2 forces it everywhere, 1 nowhere, 3 keeps the pair form.
        // Rows of a few entries (CSR feature matrices: 5-18 per row) never fill the four-neighbour steps: 24 -> 35 us for
        // the Penn94-shaped feature projection, so the vector form also needs a mean row length of 16.
        // (iii) gathered tables that fit the L2 (Squirrel / Chameleon / Cora sizes) take it for any channel count: there
        // the rows arrive at L2 speed and the instruction count is what bounds the kernel (Squirrel with the structure
        // channel: conv_bwd_spmm 57.8 -> 42.6 us, conv_fwd 64.1 -> 57.7, step 0.283 -> 0.265 ms; it replaces the
        // two-neighbours-per-instruction pair form of round 1 on those graphs).
        const bool l2_resident = (size_t)a->n_cols * F * NG * sizeof(float) <= (8u << 20);
        const int form = acm_tuning().wide_form;
        bool vec = !bf16 && F % 4 == 0 && form != 1 &&
                   ((NG == 1 && a->nnz >= 16 * a->n_rows) || (l2_resident && NG > 1 && a->nnz >= 4 * a->n_rows) || form == 2);
        for (int c = 0; c < NG && vec; ++c)
            vec = ((uintptr_t)g.p[c]) % 16 == 0 && g.ld[c] % 4 == 0 &&
                  (uint64_t)a->n_cols * (uint64_t)g.ld[c] * 4u < (1ull << 32);
        if (vec && form != 3) pair32 = false;
        // bf16 tables (even 8 < F <= 64): the vector form with 8-byte fetches whenever the fp32 operand would take it (rows of
        // 4 k columns, 8-byte aligned); the two-neighbours-per-instruction pair kernel otherwise.  On the twitch-shaped
        // graph the pair kernel is SLOWER than the fp32 vector form (conv_bwd_spmm 619 -> 707 us: half the bytes, but two
        // neighbours per instruction instead of four)
        // ... except under the fused head once the tables outgrow the 256 MB Infinity Cache: every row then comes from
        // HBM, the kernel lives on loads in flight, and the vector layout (the head four times, fewer waves) loses to the
        // pair kernel -- pokec-shaped forward (1.63 M rows, 418 MB of bf16 tables) 5.39 -> 3.89 ms, while the head-less
        // transposed gather of the backward keeps the vector form (2.32 against 2.88 ms): profiles/r04_bench_scale.jsonl
        const bool head_beyond_cache = __is_same(Epi, EpiFwd) && (size_t)a->n_cols * F * NG * 2u > ((size_t)256 << 20);
        bool vec16 = bf16 && F % 4 == 0 && form != 1 && form != 3 &&
                     ((NG == 1 && a->nnz >= 16 * a->n_rows) || (NG > 1 && a->nnz >= 4 * a->n_rows && !head_beyond_cache) ||
                      form == 2);
        for (int c = 0; c < NG && vec16; ++c)
            vec16 = ((uintptr_t)g.p[c]) % 8 == 0 && g.ld[c] % 4 == 0 && (uint64_t)a->n_cols * (uint64_t)g.ld[c] * 2u < (1ull << 32);
        if (vec16) {
            hipLaunchKernelGGL((spmm_vec_kernel<NG, 1, Epi, true>), dim3(grid), dim3(256), 0, st, v, g, F, ea, partial);
        } else if (vec && !pair32) {
            if (F <= 64)
                hipLaunchKernelGGL((spmm_vec_kernel<NG, 1, Epi>), dim3(grid), dim3(256), 0, st, v, g, F, ea, partial);
            else if (F <= 128)
                hipLaunchKernelGGL((spmm_vec_kernel<NG, 2, Epi>), dim3(grid), dim3(256), 0, st, v, g, F, ea, partial);
            else if (F <= 192)
                hipLaunchKernelGGL((spmm_vec_kernel<NG, 3, Epi>), dim3(grid), dim3(256), 0, st, v, g, F, ea, partial);
            else
                hipLaunchKernelGGL((spmm_vec_kernel<NG, 4, Epi>), dim3(grid), dim3(256), 0, st, v, g, F, ea, partial);
        } else if (bf16)
            hipLaunchKernelGGL((spmm_pair_kernel<NG, Epi, true>), dim3(grid), dim3(256), 0, st, v, g, F, ea, partial);
        else if (pair32)
            hipLaunchKernelGGL((spmm_pair_kernel<NG, Epi, false>), dim3(grid), dim3(256), 0, st, v, g, F, ea, partial);
        else if (F <= 64)
            hipLaunchKernelGGL((spmm_wide_kernel<1, NG, Epi>), dim3(grid), dim3(256), 0, st, v, g, F, ea, partial);
        else if (F <= 128)
            hipLaunchKernelGGL((spmm_wide_kernel<2, NG, Epi>), dim3(grid), dim3(256), 0, st, v, g, F, ea, partial);
        else
            hipLaunchKernelGGL((spmm_wide_kernel<4, NG, Epi>), dim3(grid), dim3(256), 0, st, v, g, F, ea, partial);
    }
    ACM_CHECK_HIP(hipGetLastError());
    if (defer_fixup)                        // the caller's next kernel adds the partial slots of the long rows itself -- except
        return (F <= 8 && narrow_finishes_long_rows(a)) ? finish_window_rows<NG, Epi>(a, v, F, ea, partial, st) : ACM_OK;   // where
                                            // the gather finishes them (sixteen lanes per item): then also the rows of several windows
    if (a->n_long && F <= 8) {
        if (narrow_finishes_long_rows(a)) return finish_window_rows<NG, Epi>(a, v, F, ea, partial, st);
        const int grid = (int)((a->n_long + 15) / 16);
        if (F <= 2)
            hipLaunchKernelGGL((spmm_fixup_narrow_kernel<2, NG, Epi>), dim3(grid), dim3(256), 0, st, v, F, ea, partial);
        else if (F <= 4)
            hipLaunchKernelGGL((spmm_fixup_narrow_kernel<4, NG, Epi>), dim3(grid), dim3(256), 0, st, v, F, ea, partial);
        else
            hipLaunchKernelGGL((spmm_fixup_narrow_kernel<8, NG, Epi>), dim3(grid), dim3(256), 0, st, v, F, ea, partial);
        ACM_CHECK_HIP(hipGetLastError());
    } else if (a->n_long) {
        ACM_REQUIRE(F <= 256, ACM_EUNSUPPORTED, "%s: F = %d > 256", who, F);
        const int grid = (int)((a->n_long + 3) / 4);
        if (F <= 64)
            hipLaunchKernelGGL((spmm_fixup_kernel<1, NG, Epi>), dim3(grid), dim3(256), 0, st, v, F, ea, partial);
        else if (F <= 128)
            hipLaunchKernelGGL((spmm_fixup_kernel<2, NG, Epi>), dim3(grid), dim3(256), 0, st, v, F, ea, partial);
        else
            hipLaunchKernelGGL((spmm_fixup_kernel<4, NG, Epi>), dim3(grid), dim3(256), 0, st, v, F, ea, partial);
        ACM_CHECK_HIP(hipGetLastError());
    }
    return ACM_OK;
}

}  // namespace

__global__ __launch_bounds__(256) void cast_bf16_kernel(long n_rows, int n_cols, const float* __restrict__ src, long ld_src,
                                                        unsigned short* __restrict__ dst, long ld_dst) {
    const long total = n_rows * n_cols;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long r = q / n_cols;
        const int c = (int)(q - r * n_cols);
        unsigned b = __float_as_uint(src[r * ld_src + c]);
        b += 0x7FFFu + ((b >> 16) & 1u);                 // round to nearest even
        dst[r * ld_dst + c] = (unsigned short)(b >> 16);
    }
}

extern "C" int acm_cast_bf16(int64_t n_rows, int64_t n_cols, const float* src, int64_t ld_src, uint16_t* dst,
                             int64_t ld_dst, acm_stream_t stream) {
    ACM_REQUIRE(src && dst, ACM_EINVAL, "acm_cast_bf16: NULL pointer");
    ACM_REQUIRE(n_rows >= 0 && n_cols >= 0 && n_cols < INT32_MAX && ld_src >= n_cols && ld_dst >= n_cols, ACM_ESHAPE,
                "acm_cast_bf16: bad sizes");
    if (n_rows == 0 || n_cols == 0) return ACM_OK;
    long blocks = (n_rows * n_cols + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long)n_rows,
                       (int)n_cols, src, (long)ld_src, dst, (long)ld_dst);
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}

extern "C" int acm_conv_head_fwd(int64_t n_rows, const acm_conv_fwd_t* p, acm_stream_t stream) {
    ACM_REQUIRE(p, ACM_EINVAL, "acm_conv_head_fwd: NULL argument");
    ACM_REQUIRE(n_rows >= 0 && n_rows < INT32_MAX, ACM_ESHAPE, "acm_conv_head_fwd: bad row count");
    ACM_REQUIRE(p->f_out == 64 && p->n_channels == 3 && !p->gather_bf16 && !p->row_scale && !p->deg, ACM_EUNSUPPORTED,
                "acm_conv_head_fwd: three fp32 channels of 64 columns, no row scale (got F %d, k %d)", p->f_out, p->n_channels);
    ACM_REQUIRE(p->g_low && p->g_high && p->s_high && p->s_mlp && p->out && p->pre && p->att && p->att_mix, ACM_EINVAL,
                "acm_conv_head_fwd: NULL pointer in the parameter block");
    const uintptr_t bits = (uintptr_t)p->g_low | (uintptr_t)p->g_high | (uintptr_t)p->s_high | (uintptr_t)p->s_mlp | (uintptr_t)p->out |
                           (uintptr_t)p->pre | (uintptr_t)p->att;
    ACM_REQUIRE((p->ld_g_low | p->ld_g_high | p->ld_s_high | p->ld_s_mlp | p->ld_out | p->ld_pre) % 4 == 0 && (bits & 15) == 0,
                ACM_EUNSUPPORTED, "acm_conv_head_fwd: every row (inputs, out, pre, att) must be 16-byte aligned");
    if (n_rows == 0) return ACM_OK;
    hipLaunchKernelGGL(conv_head_rows_kernel, dim3((unsigned)((n_rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, *p, (int)n_rows);
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}

extern "C" int acm_spmm_ex(const acm_csr_t* a, const void* G, int64_t ldg, int width, float* Y, int64_t ldy,
                           const acm_spmm_opts_t* o, void* workspace, size_t workspace_bytes, acm_stream_t stream) {
    return acm_spmm_internal(a, G, ldg, width, Y, ldy, o, workspace, workspace_bytes, stream, nullptr);
}

// *defer_fixup (in): leave the partial sums of the long rows in the workspace slots (width <= 256) for the caller's
// next kernel to add (acm_conv_agg_fwd's epilogue); (out): false when the gather finished those rows itself (the narrow
// kernel with 16 lanes per item) or the width rules it out -- Y is complete then.
int acm_spmm_internal(const acm_csr* a, const void* G, int64_t ldg, int width, float* Y, int64_t ldy,
                      const acm_spmm_opts_t* o, void* workspace, size_t workspace_bytes, acm_stream_t stream,
                      bool* defer_fixup) {
    bool defer = defer_fixup && *defer_fixup && a && width <= 256 && !(width <= 8 && narrow_finishes_long_rows(a));
    if (defer_fixup) *defer_fixup = defer;
    static const acm_spmm_opts_t none = {nullptr, nullptr, nullptr, 0, nullptr, 0, 0};
    if (!o) o = &none;
    ACM_REQUIRE(a && G && Y, ACM_EINVAL, "acm_spmm: NULL argument");
    ACM_REQUIRE(width > 0 && ldg >= width && ldy >= width && (!o->sub || o->ld_sub >= width), ACM_ESHAPE,
                "acm_spmm: width %d ldg %lld ldy %lld ld_sub %lld", width, (long long)ldg, (long long)ldy,
                (long long)o->ld_sub);
    ACM_REQUIRE(!o->g_bf16 || width <= 256, ACM_EUNSUPPORTED, "acm_spmm: bf16 operands are one column block wide");
    for (int c0 = 0; c0 < width; c0 += 256) {  // column blocks of <= 256
        const int wd = width - c0 < 256 ? width - c0 : 256;
        GatherSrc g = {{reinterpret_cast<const float*>(G) + (o->g_bf16 ? 0 : c0), nullptr, nullptr}, {ldg, 0, 0}};
        EpiPlain::Args ea = {Y + c0, ldy, o->relu, o->sub ? o->sub + c0 : nullptr, o->ld_sub, o->sub_scale, o->row_scale};
        int st = launch_gather<1, EpiPlain>(a, g, wd, ea, workspace, workspace_bytes, (hipStream_t)stream, "acm_spmm",
                                            o->vals, o->g_bf16 != 0, defer);
        if (st != ACM_OK) return st;
    }
    return ACM_OK;
}

extern "C" int acm_spmm_v(const acm_csr_t* a, const float* vals, const float* G, int64_t ldg, int width, float* Y,
                          int64_t ldy, int relu, void* workspace, size_t workspace_bytes, acm_stream_t stream) {
    const acm_spmm_opts_t o = {vals, nullptr, nullptr, 0, nullptr, relu, 0};
    return acm_spmm_ex(a, G, ldg, width, Y, ldy, &o, workspace, workspace_bytes, stream);
}

extern "C" int acm_spmm(const acm_csr_t* a, const float* G, int64_t ldg, int width, float* Y,
                        int64_t ldy, void* workspace, size_t workspace_bytes, acm_stream_t stream) {
    return acm_spmm_ex(a, G, ldg, width, Y, ldy, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int acm_conv_fwd(const acm_csr_t* a, const acm_conv_fwd_t* p, void* workspace,
                            size_t workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(a && p, ACM_EINVAL, "acm_conv_fwd: NULL argument");
    const int F = p->f_out, k = p->n_channels;
    ACM_REQUIRE(F > 0 && (k == 3 || k == 4), ACM_ESHAPE, "acm_conv_fwd: f_out %d n_channels %d", F, k);
    ACM_REQUIRE(p->g_low && p->g_high && p->s_high && p->s_mlp && p->out && p->pre && p->att &&
                    p->att_mix, ACM_EINVAL, "acm_conv_fwd: NULL tensor pointer");
    for (int c = 0; c < k; ++c) {
        ACM_REQUIRE(p->att_vec[c], ACM_EINVAL, "acm_conv_fwd: att_vec[%d] is NULL", c);
        ACM_REQUIRE(!p->layernorm || (p->ln_weight[c] && p->ln_bias[c]), ACM_EINVAL,
                    "acm_conv_fwd: layernorm parameters of channel %d are NULL", c);
    }
    ACM_REQUIRE(p->ld_out >= F && p->ld_pre >= (k - 1) * F, ACM_ESHAPE,
                "acm_conv_fwd: ld_out %lld / ld_pre %lld too small", (long long)p->ld_out,
                (long long)p->ld_pre);
    ACM_REQUIRE(((uintptr_t)p->att) % 16 == 0, ACM_EINVAL, "acm_conv_fwd: att must be 16-byte aligned");
    ACM_REQUIRE(k == 3 || (p->g_struc && p->s_struc && p->deg), ACM_EINVAL,
                "acm_conv_fwd: structure channel pointers are NULL");
    if (F <= 8) {                       // two phases: gather raw sums into `pre`, then one thread per row
        ACM_REQUIRE(!p->gather_bf16, ACM_EUNSUPPORTED, "acm_conv_fwd: bf16 operands need F > 8");
        hipStream_t s = (hipStream_t)stream;
        int st;
        if (k == 4) {
            GatherSrc g = {{p->g_low, p->g_high, p->g_struc}, {p->ld_g_low, p->ld_g_high, p->ld_g_struc}};
            st = launch_gather<3, EpiRaw>(a, g, F, *p, workspace, workspace_bytes, s, "acm_conv_fwd", nullptr, false, true);
        } else {
            GatherSrc g = {{p->g_low, p->g_high, nullptr}, {p->ld_g_low, p->ld_g_high, 0}};
            st = launch_gather<2, EpiRaw>(a, g, F, *p, workspace, workspace_bytes, s, "acm_conv_fwd", nullptr, false, true);
        }
        if (st != ACM_OK || a->n_rows == 0) return st;
        const int grid = (int)((a->n_rows + 255) / 256), n = (int)a->n_rows;
        const int FP = F <= 2 ? 2 : (F <= 4 ? 4 : 8);
        CsrView cv = acm_view(a);
        const float* part = (const float*)workspace;
        const bool done = narrow_finishes_long_rows(a);     // the gather left complete raw sums for every row
        if (done) cv.long_index = nullptr;
        const int tail = done ? 0 : (int)((a->n_long + 15) / 16);
#define ACM_ROWS(FPv)                                                                              \
    do {                                                                                           \
        if (k == 4)                                                                                                   \
            hipLaunchKernelGGL((conv_fwd_rows_kernel<FPv, 3>), dim3(grid + tail), dim3(256), 0, s, *p, n, cv, part, grid); \
        else                                                                                                          \
            hipLaunchKernelGGL((conv_fwd_rows_kernel<FPv, 2>), dim3(grid + tail), dim3(256), 0, s, *p, n, cv, part, grid); \
    } while (0)
        if (FP == 2) ACM_ROWS(2);
        else if (FP == 4) ACM_ROWS(4);
        else ACM_ROWS(8);
#undef ACM_ROWS
        ACM_CHECK_HIP(hipGetLastError());
        return ACM_OK;
    }
    if (k == 4) {
        GatherSrc g = {{p->g_low, p->g_high, p->g_struc}, {p->ld_g_low, p->ld_g_high, p->ld_g_struc}};
        return launch_gather<3, EpiFwd>(a, g, F, *p, workspace, workspace_bytes, (hipStream_t)stream,
                                        "acm_conv_fwd", nullptr, p->gather_bf16 != 0);
    }
    GatherSrc g = {{p->g_low, p->g_high, nullptr}, {p->ld_g_low, p->ld_g_high, 0}};
    return launch_gather<2, EpiFwd>(a, g, F, *p, workspace, workspace_bytes, (hipStream_t)stream,
                                    "acm_conv_fwd", nullptr, p->gather_bf16 != 0);
}

extern "C" int acm_conv_bwd_spmm(const acm_csr_t* at, const acm_conv_bwd_spmm_t* p, void* workspace,
                                 size_t workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(at && p, ACM_EINVAL, "acm_conv_bwd_spmm: NULL argument");
    const int F = p->f_out;
    ACM_REQUIRE(F > 0, ACM_ESHAPE, "acm_conv_bwd_spmm: f_out %d", F);
    ACM_REQUIRE(p->g_low && p->g_high && p->s_high && p->dz_low && p->dz_high, ACM_EINVAL,
                "acm_conv_bwd_spmm: NULL tensor pointer");
    if (p->g_struc) ACM_REQUIRE(p->s_struc && p->d_struc, ACM_EINVAL, "acm_conv_bwd_spmm: structure channel pointers are NULL");
    // wide layers on graphs whose gathered tables exceed the L2: one channel per pass (see EpiBwdLow)
    // (acm_tuning_t.bwd_split = 1 / 0 force either form, for tests and A/B measurements)
    // Measured (profiles/r02_wide_kernels.jsonl): twitch-shaped (mean degree 82) 640 -> 613 us, with the structure channel
    // 1004 -> 899, Penn94-shaped (66) 121 -> 106; arXiv-year-shaped (15) 160 -> 190: short rows pay the per-item cost of
    // every pass, so the split needs a mean degree of 32.
    const bool big = (size_t)at->n_cols * (size_t)F * sizeof(float) > (8u << 20) && at->nnz >= 32 * at->n_rows;
    // bf16 tables (gather_bf16): [G_L | G_H] of a neighbour are 2 x 128 bytes -- what ONE fp32 channel is -- so the fused pass
    // keeps the hot set of a single fp32 pass and saves the second walk over the operator (twitch-shaped, F = 64: 430 us in
    // two passes, 387 us fused; fp32: 619 us in two passes)
    const bool b16 = p->gather_bf16 != 0;             // launch_gather checks the shape (even 8 < F <= 64) and alignment
    const int want_split = acm_tuning().bwd_split;
    const bool split = want_split == 0 ? false : (want_split == 1 ? true : (big && !b16));
    if (F > 8 && F <= 256 && split) {
        hipStream_t s = (hipStream_t)stream;
        GatherSrc gl = {{p->g_low, nullptr, nullptr}, {p->ld_g_low, 0, 0}};
        int st = launch_gather<1, EpiBwdLow>(at, gl, F, *p, workspace, workspace_bytes, s, "acm_conv_bwd_spmm", nullptr, b16);
        if (st != ACM_OK) return st;
        GatherSrc gh = {{p->g_high, nullptr, nullptr}, {p->ld_g_high, 0, 0}};
        st = launch_gather<1, EpiBwdHigh>(at, gh, F, *p, workspace, workspace_bytes, s, "acm_conv_bwd_spmm", nullptr, b16);
        if (st != ACM_OK || !p->g_struc) return st;
        GatherSrc gs = {{p->g_struc, nullptr, nullptr}, {p->ld_g_struc, 0, 0}};
        return launch_gather<1, EpiBwdStruc>(at, gs, F, *p, workspace, workspace_bytes, s, "acm_conv_bwd_spmm", nullptr, b16);
    }
    if (p->g_struc) {
        GatherSrc g = {{p->g_low, p->g_high, p->g_struc}, {p->ld_g_low, p->ld_g_high, p->ld_g_struc}};
        return launch_gather<3, EpiBwd>(at, g, F, *p, workspace, workspace_bytes, (hipStream_t)stream,
                                        "acm_conv_bwd_spmm", nullptr, b16);
    }
    GatherSrc g = {{p->g_low, p->g_high, nullptr}, {p->ld_g_low, p->ld_g_high, 0}};
    return launch_gather<2, EpiBwd>(at, g, F, *p, workspace, workspace_bytes, (hipStream_t)stream,
                                    "acm_conv_bwd_spmm", nullptr, b16);
}

// ================================================================== K3: row-local backward
// Parameter-gradient vector layout (npg = 3 k F + k k floats):
//   [ d att_vec : k x F ][ d ln_weight : k x F ][ d ln_bias : k x F ][ d att_mix : k x k ]
template <class L, int K>
__device__ __forceinline__ void conv_bwd_row(const acm_conv_bwd_local_t& p, int row, bool active,
                                             const L& lay, ParamAcc<L>& pa) {
    constexpr int NV = L::NV;
    constexpr int k = K;
    const int F = p.f_out;
    float H[4][NV], hn[4][NV], xhat[4][NV], dO[NV];
    bool pos[4][NV];
    const float dg = (k == 4 && active && p.deg) ? p.deg[row] : 1.f;
    const float gsc = (active && p.g_scale) ? p.g_scale[row] : 1.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = lay.col(i);
        const bool ok = active && col < F;
        const float* pr = p.pre + (long)row * p.ld_pre;
        const float p0 = ok ? pr[col] : 0.f;
        const float p1 = ok ? pr[F + col] : 0.f;
        const float p3 = (ok && k == 4) ? pr[2 * F + col] : 0.f;
        const float zi = ok ? p.s_mlp[(long)row * p.ld_s_mlp + col] : 0.f;
        dO[i] = ok ? p.grad_out[(long)row * p.ld_grad_out + col] : 0.f;
        pos[0][i] = p.relu_after ? (p0 > 0.f) : true;
        pos[1][i] = p.relu_after ? (p1 > 0.f) : true;
        pos[2][i] = p.relu_mlp ? (zi > 0.f) : true;
        pos[3][i] = p3 > 0.f;
        H[0][i] = pos[0][i] ? p0 : 0.f;
        H[1][i] = pos[1][i] ? p1 : 0.f;
        H[2][i] = pos[2][i] ? zi : 0.f;
        H[3][i] = pos[3][i] ? p3 : 0.f;
    }
    HeadOut ho;
    const HeadParams hp = acm_head_params(p);
    acm_head<L, K>(lay, F, p.layernorm, hp, H, hn, xhat, ho);

    if (p.post_relu || p.post_scale || p.post_drop.p > 0.f) {   // undo the fused post-op of the forward on the incoming gradient
        const AcmDropCtx dc = acm_drop_ctx(p.post_drop);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = lay.col(i);
            const bool ok = active && col < F;
            float raw = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < K) raw += ho.alpha[c] * H[c][i];
            if (p.post_relu && !(raw * p.scale > 0.f)) dO[i] = 0.f;
            if (p.post_scale && ok) dO[i] *= p.post_scale[(long)row * p.ld_post_scale + col];
            if (dc.on && ok) dO[i] *= acm_drop1(dc, row, col);
        }
    }
    float dH[4][NV];
    acm_head_backward<L, K>(lay, F, p.layernorm, hp, p.scale, H, hn, xhat, ho, dO, active ? 1.f : 0.f, pa, dH);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= k) continue;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = lay.col(i);
            if (!(active && col < F)) continue;
            const float gval = pos[c][i] ? dH[c][i] : 0.f;
            if (c == 0) p.g_low[(long)row * p.ld_g_low + col] = gsc * gval;
            if (c == 1) p.g_high[(long)row * p.ld_g_high + col] = gsc * gval;
            if (c == 2) p.g_mlp[(long)row * p.ld_g_mlp + col] = gval;
            if (c == 3) p.g_struc[(long)row * p.ld_g_struc + col] = dg * gval;
        }
    }
}

// Block-level deterministic reduction of the per-lane parameter-gradient accumulators of K3:
// the RPW row-groups of a wave (shuffles) -> LDS slab [4 waves][npg] -> sum over the waves -> out[npg].
template <class L, int RPW, int K>
__device__ __forceinline__ void bwd_local_block_reduce(ParamAcc<L>& pa, const L& lay, int F, float* lds,
                                                       float* __restrict__ out) {
    constexpr int k = K;
    const int npg = 3 * k * F + k * k;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // combine the RPW row-groups of this wave.  One row per lane (RPW = 64: the thread-per-row kernels of the narrow
    // layers) sums all 64 lanes with the DPP / permlane tree of acm_group_sum -- the xor butterfly below lowers to one
    // ds_bpermute + s_waitcnt per step, 6 steps x (12 NV + 16) values per block: it was most of the fused output-layer
    // tail's 18 us
    if (RPW == 64) {
#pragma unroll
        for (int c = 0; c < k; ++c)
#pragma unroll
            for (int i = 0; i < L::NV; ++i) {
                pa.dv[c][i] = acm_group_sum<64>(pa.dv[c][i]);
                pa.dgam[c][i] = acm_group_sum<64>(pa.dgam[c][i]);
                pa.dbet[c][i] = acm_group_sum<64>(pa.dbet[c][i]);
            }
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if ((q >> 2) < k && (q & 3) < k) pa.dmix[q] = acm_group_sum<64>(pa.dmix[q]);
    } else if (RPW == 8) {                       // eight lanes per row: lanes l, l ^ 8 (DPP row_ror:8), then the four 16-lane rows
#pragma unroll
        for (int c = 0; c < k; ++c)
#pragma unroll
            for (int i = 0; i < L::NV; ++i) {
                pa.dv[c][i] = acm_cross_row_sum(pa.dv[c][i] + acm_dpp<0x128>(pa.dv[c][i]));
                pa.dgam[c][i] = acm_cross_row_sum(pa.dgam[c][i] + acm_dpp<0x128>(pa.dgam[c][i]));
                pa.dbet[c][i] = acm_cross_row_sum(pa.dbet[c][i] + acm_dpp<0x128>(pa.dbet[c][i]));
            }
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if ((q >> 2) < k && (q & 3) < k) pa.dmix[q] = acm_cross_row_sum(pa.dmix[q] + acm_dpp<0x128>(pa.dmix[q]));
    } else if (RPW > 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < L::NV; ++i) {
                for (int m = 64 / RPW; m < 64; m <<= 1) {
                    pa.dv[c][i] += __shfl_xor(pa.dv[c][i], m, 64);
                    pa.dgam[c][i] += __shfl_xor(pa.dgam[c][i], m, 64);
                    pa.dbet[c][i] += __shfl_xor(pa.dbet[c][i], m, 64);
                }
            }
#pragma unroll
        for (int q = 0; q < 16; ++q)
            for (int m = 64 / RPW; m < 64; m <<= 1) pa.dmix[q] += __shfl_xor(pa.dmix[q], m, 64);
    }
    float* slab = lds + wv * npg;
    const bool writer = (RPW == 1) || (lane < 64 / RPW);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= k) continue;
#pragma unroll
        for (int i = 0; i < L::NV; ++i) {
            const int col = lay.col(i);
            if (writer && col < F) {
                slab[(0 * k + c) * F + col] = pa.dv[c][i];
                slab[(1 * k + c) * F + col] = pa.dgam[c][i];
                slab[(2 * k + c) * F + col] = pa.dbet[c][i];
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c < k && j < k) slab[3 * k * F + c * k + j] = pa.dmix[c * 4 + j];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < npg; q += 256)
        out[q] = (lds[q] + lds[npg + q]) + (lds[2 * npg + q] + lds[3 * npg + q]);
}


// K3, one launch: rows -> accumulators -> bwd_local_block_reduce -> partial[block][npg].
template <class L, int RPW /* rows per wave */, int K>
__global__ __launch_bounds__(256) void conv_bwd_local_kernel(acm_conv_bwd_local_t p, int n_rows,
                                                             float* __restrict__ partial) {
    extern __shared__ float lds[];
    constexpr int k = K;
    const int F = p.f_out;
    const int npg = 3 * k * F + k * k;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    L lay{lane};
    ParamAcc<L> pa;
    pa.zero();
    const int rows_per_block = 4 * RPW;
    for (int r0 = blockIdx.x * rows_per_block; r0 < n_rows; r0 += gridDim.x * rows_per_block) {
        const int row = r0 + wv * RPW + (RPW > 1 ? lane / (64 / RPW) : 0);
        conv_bwd_row<L, K>(p, row < n_rows ? row : 0, row < n_rows, lay, pa);
    }
    bwd_local_block_reduce<L, RPW, K>(pa, lay, F, lds, partial + (long)blockIdx.x * npg);
}

// K3 for 16 < F <= 64: four rows per wave (16 lanes x 4 columns), head parameters in LDS, two passes per
// row (scalars, then one channel at a time).  Same partial-vector layout as conv_bwd_local_kernel, so the
// same reduce kernel finishes the job.  (The one-row-per-wave version spent 270-370 us here on the
// twitch-sized graph: every lane recomputed the row scalars and the compiler parked the loop-invariant
// parameter loads in ~36 VGPRs.)
template <int K>
__global__ __launch_bounds__(256) void conv_bwd_local_grouped_kernel(acm_conv_bwd_local_t p, int n_rows,
                                                                     float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    const int F = p.f_out;
    const int npg = 3 * K * F + K * K;
    float* hlds = lds;                                   // 3 * K * 64 floats, dead after the row loop
    stage_head_params<K>(hlds, p.att_vec, p.ln_weight, p.ln_bias, p.layernorm, F);
    __syncthreads();
    float pA[K][4], pS[K], dmix1 = 0.f, mixm[K * K];     // head-parameter accumulators (see row_channel_backward)
    const int qc = (m < K * K ? m : 0) / K, qj = (m < K * K ? m : 0) % K;    // the att_mix element this lane accumulates
#pragma unroll
    for (int c = 0; c < K; ++c) {
        pS[c] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) pA[c][i] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < K * K; ++q) {
        mixm[q] = p.att_mix[q];
    }
    const bool ln = p.layernorm != 0;
    for (int r0 = (blockIdx.x * 4 + wv) * 4; r0 < n_rows; r0 += gridDim.x * 16) {
        const int row = r0 + g;
        const bool active = row < n_rows;
        const long rr = active ? row : 0;
        // 32-bit element offsets from the (uniform) base pointers: one VGPR per array instead of a
        // loop-carried 64-bit pointer per access (the host checks n_rows * ld < 2^31)
        const unsigned urow = (unsigned)rr;
        const int mm = acm_opaque(m);             // see acm_opaque(): keeps the LDS parameter reads in the loop
        float H[K][4], dO[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = m + 16 * i;
            const bool ok = active && col < F;
            const unsigned cc = ok ? (unsigned)col : 0u;          // clamped: loads stay unconditional (no exec branches)
            const unsigned o_pre = urow * (unsigned)p.ld_pre + cc;
            const float p0 = p.pre[o_pre], p1 = p.pre[o_pre + F];
            const float zi = p.s_mlp[urow * (unsigned)p.ld_s_mlp + cc];
            const float go = p.grad_out[urow * (unsigned)p.ld_grad_out + cc];
            H[0][i] = ok ? (p.relu_after ? fmaxf(p0, 0.f) : p0) : 0.f;
            H[1][i] = ok ? (p.relu_after ? fmaxf(p1, 0.f) : p1) : 0.f;
            H[2][i] = ok ? (p.relu_mlp ? fmaxf(zi, 0.f) : zi) : 0.f;
            if (K == 4) H[K - 1][i] = ok ? fmaxf(p.pre[o_pre + 2 * F], 0.f) : 0.f;
            dO[i] = ok ? go : 0.f;
        }
        RowHead<K> rh;
        row_head<K>(hlds, mixm, mm, F, ln, H, rh);
        row_post_backward<K>(p, rh, H, active, rr, m, F, dO);
        float ds[K];
        row_head_backward_scalars<K>(rh, mixm, p.scale, H, dO, ds, qc, qj, dmix1);
        const float dg = (K == 4 && active && p.deg) ? p.deg[rr] : 1.f;
        const float gsc = (active && p.g_scale) ? p.g_scale[rr] : 1.f;
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const bool relu_c = (c < 2) ? (p.relu_after != 0) : (c == 2 ? p.relu_mlp != 0 : true);
            float G[4];
            row_channel_backward<K>(hlds, c, mm, F, ln, p.scale, rh, ds[c], H[c], dO, pA[c], pS[c], G);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = m + 16 * i;
                if (!(active && col < F)) continue;
                const float gv = (!relu_c || H[c][i] > 0.f) ? G[i] : 0.f;
                if (c == 0) p.g_low[urow * (unsigned)p.ld_g_low + col] = gsc * gv;
                if (c == 1) p.g_high[urow * (unsigned)p.ld_g_high + col] = gsc * gv;
                if (c == 2) p.g_mlp[urow * (unsigned)p.ld_g_mlp + col] = gv;
                if (c == 3) p.g_struc[urow * (unsigned)p.ld_g_struc + col] = dg * gv;
            }
        }
    }
    float dv[K][4], dgam[K][4], dbet[K][4];
#pragma unroll
    for (int c = 0; c < K; ++c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pA[c][i] = acm_cross_row_sum(pA[c][i]);
        pS[c] = acm_cross_row_sum(pS[c]);
        row_param_grads<K>(hlds, c, m, pA[c], pS[c], dv[c], dgam[c], dbet[c]);      // hlds is still intact here
    }
    dmix1 = acm_cross_row_sum(dmix1);
    __syncthreads();
    float* slab = lds + wv * npg;
    if (g == 0) {
#pragma unroll
        for (int c = 0; c < K; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = m + 16 * i;
                if (col < F) {
                    slab[(0 * K + c) * F + col] = dv[c][i];
                    slab[(1 * K + c) * F + col] = dgam[c][i];
                    slab[(2 * K + c) * F + col] = dbet[c][i];
                }
            }
    }
    if (g == 0 && m < K * K) slab[3 * K * F + m] = dmix1;
    __syncthreads();
    for (int q = threadIdx.x; q < npg; q += 256)
        partial[(long)blockIdx.x * npg + q] = (lds[q] + lds[npg + q]) + (lds[2 * npg + q] + lds[3 * npg + q]);
}

namespace {
int bwd_local_blocks(int64_t n_rows, int rows_per_block) {
    int64_t nb = (n_rows + rows_per_block - 1) / rows_per_block;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// Second phase of K3: the [d att_vec | d ln_weight | d ln_bias] (k x F each) | d att_mix (k x k) columns of the
// per-block partials go to up to 3k + 1 destinations.
int bwd_local_reduce(const acm_conv_bwd_local_t* p, const float* partial, int nblk, hipStream_t st) {
    const int F = p->f_out, k = p->n_channels, npg = 3 * k * F + k * k;
    acm_reduce_seg_t segs[13];
    int n = 0;
    for (int which = 0; which < 3; ++which)
        for (int c = 0; c < k; ++c) {
            float* dst = which == 0 ? p->d_att_vec[c] : (which == 1 ? p->d_ln_weight[c] : p->d_ln_bias[c]);
            if (dst) segs[n++] = {partial, nblk, npg, (which * k + c) * F, F, dst, F, 0, 0, 0};
        }
    segs[n++] = {partial, nblk, npg, 3 * k * F, k * k, p->d_att_mix, k * k, 0, 0, 0};
    return acm_reduce_emit(p->defer, segs, n, st);
}
int bwd_rows_per_wave(int F) { return F > 64 ? 1 : (F > 16 ? 4 : (F > 8 ? 4 : (F > 4 ? 8 : (F > 2 ? 16 : 32)))); }
}  // namespace

extern "C" int acm_conv_bwd_local_workspace_bytes(int64_t n_rows, int f_out, int n_channels, size_t* bytes) {
    ACM_REQUIRE(bytes, ACM_EINVAL, "acm_conv_bwd_local_workspace_bytes: NULL argument");
    ACM_REQUIRE(f_out > 0 && (n_channels == 3 || n_channels == 4), ACM_ESHAPE,
                "acm_conv_bwd_local_workspace_bytes: f_out %d n_channels %d", f_out, n_channels);
    const int npg = 3 * n_channels * f_out + n_channels * n_channels;
    *bytes = (size_t)bwd_local_blocks(n_rows, 4 * bwd_rows_per_wave(f_out)) * npg * sizeof(float);
    return ACM_OK;
}

extern "C" int acm_conv_bwd_local(int64_t n_rows, const acm_conv_bwd_local_t* p, void* workspace,
                                  size_t workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(p, ACM_EINVAL, "acm_conv_bwd_local: NULL argument");
    const int F = p->f_out, k = p->n_channels;
    ACM_REQUIRE(F > 0 && F <= 256 && (k == 3 || k == 4), (F > 256 ? ACM_EUNSUPPORTED : ACM_ESHAPE),
                "acm_conv_bwd_local: f_out %d n_channels %d", F, k);
    ACM_REQUIRE(p->grad_out && p->pre && p->s_mlp && p->att_mix && p->g_low && p->g_high && p->g_mlp &&
                    p->d_att_mix, ACM_EINVAL, "acm_conv_bwd_local: NULL tensor pointer");
    ACM_REQUIRE(k == 3 || p->g_struc, ACM_EINVAL,
                "acm_conv_bwd_local: structure channel pointers are NULL");
    for (int c = 0; c < k; ++c) {
        ACM_REQUIRE(p->att_vec[c] && p->d_att_vec[c], ACM_EINVAL, "acm_conv_bwd_local: att_vec[%d] NULL", c);
        ACM_REQUIRE(!p->layernorm || (p->ln_weight[c] && p->ln_bias[c] && p->d_ln_weight[c] && p->d_ln_bias[c]),
                    ACM_EINVAL, "acm_conv_bwd_local: layernorm pointers of channel %d NULL", c);
    }
    size_t need = 0;
    acm_conv_bwd_local_workspace_bytes(n_rows, F, k, &need);
    ACM_REQUIRE(workspace && workspace_bytes >= need, ACM_ENOMEM,
                "acm_conv_bwd_local: workspace %zu B < required %zu B", workspace_bytes, need);
    const int npg = 3 * k * F + k * k;
    const int rpw = bwd_rows_per_wave(F);
    const int nblk = bwd_local_blocks(n_rows, 4 * rpw);
    size_t lds = (size_t)4 * npg * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    if (F == 64 && k == 3) {                      // sixteen rows per wave, 16-byte accesses (acm_conv_local16.hip)
        const int nb16 = acm_bwd_local16(p, n_rows, partial, nblk, st);
        if (nb16 < 0) return -nb16;
        if (nb16 > 0) return bwd_local_reduce(p, partial, nb16, st);
    }
    if (F > 16 && F <= 64) {                      // 4-rows-per-wave lean kernel
        const int64_t max_ld = p->ld_pre > p->ld_grad_out ? p->ld_pre : p->ld_grad_out;
        ACM_REQUIRE(n_rows * (max_ld > p->ld_g_mlp ? max_ld : p->ld_g_mlp) < (int64_t)INT32_MAX, ACM_EUNSUPPORTED,
                    "acm_conv_bwd_local: rows x leading dimension exceeds 2^31");
        const size_t hl = (size_t)3 * k * 64 * sizeof(float);
        if (hl > lds) lds = hl;
        if (k == 3)
            hipLaunchKernelGGL((conv_bwd_local_grouped_kernel<3>), dim3(nblk), dim3(256), lds, st, *p, (int)n_rows, partial);
        else
            hipLaunchKernelGGL((conv_bwd_local_grouped_kernel<4>), dim3(nblk), dim3(256), lds, st, *p, (int)n_rows, partial);
        ACM_CHECK_HIP(hipGetLastError());
        return bwd_local_reduce(p, partial, nblk, st);
    }
#define ACM_BWD(LAY, RPW)                                                                                   \
    do {                                                                                                    \
        if (k == 3)                                                                                         \
            hipLaunchKernelGGL((conv_bwd_local_kernel<LAY, RPW, 3>), dim3(nblk), dim3(256), lds, st, *p,    \
                               (int)n_rows, partial);                                                       \
        else                                                                                                \
            hipLaunchKernelGGL((conv_bwd_local_kernel<LAY, RPW, 4>), dim3(nblk), dim3(256), lds, st, *p,    \
                               (int)n_rows, partial);                                                       \
    } while (0)
    if (F > 128) ACM_BWD(LayWide<4>, 1);
    else if (F > 64) ACM_BWD(LayWide<2>, 1);
    else if (F > 8) ACM_BWD(LayPacked<16>, 4);
    else if (F > 4) ACM_BWD(LayPacked<8>, 8);
    else if (F > 2) ACM_BWD(LayPacked<4>, 16);
    else ACM_BWD(LayPacked<2>, 32);
#undef ACM_BWD
    ACM_CHECK_HIP(hipGetLastError());
    return bwd_local_reduce(p, partial, nblk, st);
}

// ================================================================== output layer + loss + K3 in one row pass
// One thread per row, three steps that hand their results to each other through the row's own few bytes of global
// memory (written and read back by the same thread): the head of the narrow forward, the masked NLL of its logits, the
// row-local backward with that gradient.  Block-level reductions: loss partial, K3 parameter partials.
template <int FP, int NG>
__global__ __launch_bounds__(256) void conv_tail_rows_kernel(acm_conv_fwd_t pf, acm_loss_t pl, acm_conv_bwd_local_t pb,
                                                             int n_rows, float* __restrict__ loss_partial,
                                                             float* __restrict__ k3_partial) {
    extern __shared__ float lds[];
    __shared__ float red[256];
    constexpr int K = NG + 1;
    const int F = pf.f_out;
    const int npg = 3 * K * F + K * K;
    const int row = blockIdx.x * 256 + threadIdx.x;
    const bool active = row < n_rows;
    const LaySerial<FP> lay{true};
    ParamAcc<LaySerial<FP>> pa;
    pa.zero();
    float term = 0.f;
    if (active) {
        float acc[NG][FP];
        const float* pr = pf.pre + (long)row * pf.ld_pre;
#pragma unroll
        for (int c = 0; c < NG; ++c)
#pragma unroll
            for (int f = 0; f < FP; ++f) acc[c][f] = (f < F) ? pr[c * F + f] : 0.f;
        EpiFwd::apply<LaySerial<FP>, NG>(pf, row, lay, F, acc);
        term = acm_nll_row(F, pf.out + (long)row * pf.ld_out, (int)pl.labels[row], pl.row_weight[row],
                           pl.dlogits + (long)row * pl.ld_dlogits);
        conv_bwd_row<LaySerial<FP>, K>(pb, row, true, lay, pa);      // LaySerial: no cross-lane step, divergence is fine
    }
    red[threadIdx.x] = term;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_partial[blockIdx.x] = red[0];
    bwd_local_block_reduce<LaySerial<FP>, 64, K>(pa, lay, F, lds, k3_partial + (long)blockIdx.x * npg);
}

// The same for 4 < F <= 8 with EIGHT LANES PER ROW (LayPacked<8>: one column per lane, eight rows per wave): the
// thread-per-row form above holds three channels x 8 columns of every stage in registers (169 VGPRs, 3 waves/SIMD) and
// walks a row's head, loss and backward as one serial chain -- 50 us for the 169 k rows of the arXiv-year-shaped graph.
// Here a lane reads back only what it wrote itself (its own logit, its own dlogit), the row-wise max / sums of the loss are
// 8-lane DPP reductions.
__device__ __forceinline__ float acm_group8_max(float v) {
    v = fmaxf(v, acm_dpp<0xB1>(v));      // quad_perm [1,0,3,2]
    v = fmaxf(v, acm_dpp<0x4E>(v));      // quad_perm [2,3,0,1]
    return fmaxf(v, acm_dpp<0x141>(v));  // row_half_mirror
}

template <int NG>
__global__ __launch_bounds__(256) void conv_tail_packed8_kernel(acm_conv_fwd_t pf, acm_loss_t pl, acm_conv_bwd_local_t pb,
                                                                int n_rows, float* __restrict__ loss_partial,
                                                                float* __restrict__ k3_partial) {
    extern __shared__ float lds[];
    __shared__ float red[256];
    constexpr int K = NG + 1;
    using L = LayPacked<8>;
    const int F = pf.f_out;
    const int npg = 3 * K * F + K * K;
    const int lane = threadIdx.x & 63, col = lane & 7;
    const int row = blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool active = row < n_rows;
    const int rr = active ? row : 0;
    const L lay{lane};
    ParamAcc<L> pa;
    pa.zero();
    float term = 0.f;
    {
        float acc[NG][1];
        const float* pr = pf.pre + (long)rr * pf.ld_pre;
#pragma unroll
        for (int c = 0; c < NG; ++c) acc[c][0] = (col < F) ? pr[c * F + col] : 0.f;
        if (active) EpiFwd::apply<L, NG>(pf, rr, lay, F, acc);          // every lane of an active row takes part
        // masked NLL of the row's logits: the lane's own logit back from memory (it wrote it), the rest by reduction
        const bool mine = active && col < F;
        const float z = mine ? pf.out[(long)rr * pf.ld_out + col] : -INFINITY;
        const float wi = active ? pl.row_weight[rr] : 0.f;
        const int yi = active ? (int)pl.labels[rr] : 0;
        const float m = acm_group8_max(z);
        const float e = mine ? expf(z - m) : 0.f;
        const float ssum = acm_group_sum<8>(e);
        const float zy = acm_group_sum<8>((mine && col == yi) ? z : 0.f);
        if (mine) pl.dlogits[(long)rr * pl.ld_dlogits + col] = (wi == 0.f) ? 0.f : wi * (e / ssum - (col == yi ? 1.f : 0.f));
        if (active && col == 0 && wi != 0.f) term = wi * (m + logf(ssum) - zy);
        conv_bwd_row<L, K>(pb, rr, active, lay, pa);
    }
    red[threadIdx.x] = term;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_partial[blockIdx.x] = red[0];
    bwd_local_block_reduce<L, 8, K>(pa, lay, F, lds, k3_partial + (long)blockIdx.x * npg);
}

extern "C" int acm_conv_fwd_tail_workspace_bytes(int64_t n_rows, int f_out, int n_channels, size_t* bytes) {
    ACM_REQUIRE(bytes, ACM_EINVAL, "acm_conv_fwd_tail_workspace_bytes: NULL argument");
    ACM_REQUIRE(n_rows >= 0 && f_out > 0 && f_out <= 8 && n_channels == 3, ACM_EUNSUPPORTED,
                "acm_conv_fwd_tail: f_out %d n_channels %d (needs f_out <= 8, three channels)", f_out, n_channels);
    const int64_t rows_per_block = f_out > 4 ? 32 : 256;         // 4 < F <= 8: eight lanes per row
    const int64_t nblk = (n_rows + rows_per_block - 1) / rows_per_block > 0 ? (n_rows + rows_per_block - 1) / rows_per_block : 1;
    *bytes = (size_t)nblk * (size_t)(1 + 3 * n_channels * f_out + n_channels * n_channels) * sizeof(float);
    return ACM_OK;
}

extern "C" int acm_conv_fwd_tail(const acm_csr_t* a, const acm_conv_fwd_t* p, const acm_loss_t* l,
                                 const acm_conv_bwd_local_t* b, void* workspace, size_t workspace_bytes,
                                 void* tail_workspace, size_t tail_workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(a && p && l && b, ACM_EINVAL, "acm_conv_fwd_tail: NULL argument");
    const int F = p->f_out, k = p->n_channels;
    size_t need = 0;
    int st = acm_conv_fwd_tail_workspace_bytes(a->n_rows, F, k, &need);
    if (st != ACM_OK) return st;
    ACM_REQUIRE(l->n_classes == F && b->f_out == F && b->n_channels == k, ACM_EUNSUPPORTED,
                "acm_conv_fwd_tail: the layer's f_out must be the number of classes");
    ACM_REQUIRE(!p->post_relu && !p->post_scale && p->post_drop.p == 0.f && !b->post_relu && !b->post_scale &&
                    b->post_drop.p == 0.f && !p->gather_bf16, ACM_EUNSUPPORTED,
                "acm_conv_fwd_tail: post-ops / bf16 operands are not part of the fused tail");
    ACM_REQUIRE(a->n_long == 0 || narrow_finishes_long_rows(a), ACM_EUNSUPPORTED,
                "acm_conv_fwd_tail: this graph's narrow gather leaves partial sums of long rows");
    ACM_REQUIRE(p->g_low && p->g_high && p->s_high && p->s_mlp && p->out && p->pre && p->att && p->att_mix &&
                    l->labels && l->row_weight && l->loss && l->dlogits && b->att_mix && b->g_low && b->g_high &&
                    b->g_mlp && b->d_att_mix, ACM_EINVAL, "acm_conv_fwd_tail: NULL tensor pointer");
    ACM_REQUIRE(b->grad_out == l->dlogits && b->ld_grad_out == l->ld_dlogits && b->pre == p->pre &&
                    b->ld_pre == p->ld_pre && b->s_mlp == p->s_mlp && b->ld_s_mlp == p->ld_s_mlp, ACM_EINVAL,
                "acm_conv_fwd_tail: bwd must read what fwd / loss write (grad_out = dlogits, pre, s_mlp)");
    ACM_REQUIRE(p->ld_out >= F && p->ld_pre >= (k - 1) * F && l->ld_dlogits >= F && ((uintptr_t)p->att) % 16 == 0,
                ACM_ESHAPE, "acm_conv_fwd_tail: leading dimensions / alignment");
    for (int c = 0; c < k; ++c) {
        ACM_REQUIRE(p->att_vec[c] && b->att_vec[c], ACM_EINVAL, "acm_conv_fwd_tail: att_vec[%d] is NULL", c);
        ACM_REQUIRE(!p->layernorm || (p->ln_weight[c] && p->ln_bias[c] && b->ln_weight[c] && b->ln_bias[c]), ACM_EINVAL,
                    "acm_conv_fwd_tail: layernorm parameters of channel %d are NULL", c);
    }
    ACM_REQUIRE(tail_workspace && tail_workspace_bytes >= need, ACM_ENOMEM,
                "acm_conv_fwd_tail: tail workspace %zu B < required %zu B", tail_workspace_bytes, need);
    if (a->n_rows == 0) return ACM_OK;
    hipStream_t s = (hipStream_t)stream;
    GatherSrc g = {{p->g_low, p->g_high, nullptr}, {p->ld_g_low, p->ld_g_high, 0}};
    st = launch_gather<2, EpiRaw>(a, g, F, *p, workspace, workspace_bytes, s, "acm_conv_fwd_tail", nullptr, false, true);
    if (st != ACM_OK) return st;
    const int FP = F <= 2 ? 2 : (F <= 4 ? 4 : 8);
    const bool packed = FP == 8;
    const int n = (int)a->n_rows, nblk = packed ? (n + 31) / 32 : (n + 255) / 256;
    const int npg = 3 * k * F + k * k;
    float* loss_partial = (float*)tail_workspace;
    float* k3_partial = loss_partial + nblk;
    const size_t lds = (size_t)4 * npg * sizeof(float);
    if (packed)
        hipLaunchKernelGGL((conv_tail_packed8_kernel<2>), dim3(nblk), dim3(256), lds, s, *p, *l, *b, n, loss_partial, k3_partial);
    else if (FP == 2)
        hipLaunchKernelGGL((conv_tail_rows_kernel<2, 2>), dim3(nblk), dim3(256), lds, s, *p, *l, *b, n, loss_partial, k3_partial);
    else
        hipLaunchKernelGGL((conv_tail_rows_kernel<4, 2>), dim3(nblk), dim3(256), lds, s, *p, *l, *b, n, loss_partial, k3_partial);
    ACM_CHECK_HIP(hipGetLastError());
    const acm_reduce_seg_t seg = {loss_partial, nblk, 1, 0, 1, l->loss, 1, 0, 0, 0};
    st = acm_reduce_emit(b->defer, &seg, 1, s);
    if (st != ACM_OK) return st;
    return bwd_local_reduce(b, k3_partial, nblk, s);
}
