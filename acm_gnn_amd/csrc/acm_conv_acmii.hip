// ACMII first layers with a narrow input (F_in <= 8 < F = 64): recompute on gather.
//
// ACMII (ACM-Geometric/layers.py:94-99, the default of ACM-Geometric: parse.py:57 --variant 1) puts the ReLU
// BETWEEN projection and filter,
//     H_L = A_low relu(X W_L),   H_H = relu(X W_H) - A_low relu(X W_H),   H_I = relu(X W_I),
// so the aggregate-first rewrite (A (X W) = (A X) W) is illegal and the literal form gathers the projected rows:
// 2 F floats = 512 B per edge, 7.05 GB per pass on the twitch-shaped graph, served by the Infinity Cache at
// ~7.5 TB/s (scripts/probe_wide.py) -- 714 us, plus the 50 us GEMM that wrote them.  Here the kernel gathers the
// neighbour's INPUT row instead (f_pad floats = 32 B: a 5 MB table that lives in the L2) and recomputes
// relu(x_j [W_L | W_H]) per edge on the matrix pipe:
//
//   one wave per FOUR work items; a batch = 4 neighbours of each = the 16 rows (M) of v_mfma_f32_16x16x4_f32;
//   A[i][k] = x_{j_i}[feature], lane (i = lane & 15, kq = lane >> 4) fetches the float2 (2 kq, 2 kq + 1) of the
//             neighbour in slot i & 3 of item i >> 2 (K-step s uses feature 2 kq + s: one 8-byte load per lane and batch);
//   B[k][n] = [W_L | W_H][2 kq + s][16 t + i], 16 loop-invariant registers per lane;
//   D tile t = 16 (item, neighbour) rows x 16 columns; ReLU and the sum over the tile's four row registers -- the four
//   neighbours of lane group g's item -- on the VALU, accumulated per lane: no cross-lane reduction at all.
//
// 2 x 8 x 128 FLOP per edge = 28 GFLOP per pass; at the 157 TFLOP/s fp32 MFMA peak 0.18 ms.
// Values of an explicit operator (a_ij > 0) scale the gathered row before the product: relu(a z) = a relu(z).
// The same kernel writes what the literal path's K1 GEMM would have written for the backward (the row's own
// relu(x_i [W_L | W_H | W_I]): K4's masks and self term, K3's s_mlp), so it replaces K1 + K2.
#include "acm_conv_device.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// |a| + |b| and max(x, 0) written with BUILTINS, so that the compiler sees every read of an MFMA result: its hazard
// recogniser inserts the wait states between a v_mfma and a VALU read of its VGPR results (-amdgpu-mfma-vgpr-form) only
// for instructions it scheduled itself -- reads hidden inside inline assembly are invisible to it (round 3 saw stale
// registers behind a bf16 MFMA with the inline-assembly form of this helper).  |.| folds into v_add_f32 as a source
// modifier (one instruction); the empty asm on the RESULT (no instruction, reads no MFMA register) only keeps the
// SLP vectoriser from pairing two of these into v_pk_add_f32, which has no |.| and costs a v_and_b32 per operand.
__device__ __forceinline__ float abs_add(float a, float b) {
    float r = __builtin_fabsf(a) + __builtin_fabsf(b);
    asm volatile("" : "+v"(r));
    return r;
}
// (twelve per quad of rows, outside the per-edge loop: the plain library form is good enough)
__device__ __forceinline__ float relu1(float x) { return fmaxf(x, 0.f); }

template <int K>
struct RowOut {
    float H[K][4];
};

// One wave = FOUR consecutive work items.  The 16 rows of the MFMA's A operand are (item g = row >> 2, neighbour slot
// r = row & 3); the result rows 4 g .. 4 g + 3 then sit in the four registers of lane group g, so the sum over a row's
// neighbours never crosses lanes and each 16-lane group ends up with ITS item's aggregated row in the
// 16-lane x 4-column layout the head works in (LayGrouped<4>) -- four rows per epilogue pass, no redundancy.  The
// work list is in row order (degree-sorted graphs: neighbouring items have similar lengths), so a quad's items
// finish within a batch or two of each other; idle slots multiply zeros.
template <int K>
__global__ __launch_bounds__(256) void acmii_fwd_kernel(acm_conv_acmii_fwd_t p, CsrView csr, float* __restrict__ partial) {
    constexpr int T = 8;                         // T = tiles of 16 gathered columns: [Z_L (4) | Z_H (4)], F = 64
    __shared__ __attribute__((aligned(16))) float hlds[3 * K * 64];
    const int F = 64;
    stage_head_params<K>(hlds, p.att_vec, p.ln_weight, p.ln_bias, p.layernorm, F);
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    const int ga = i >> 2, ra = i & 3;           // this lane's A-operand row: item ga of the quad, neighbour slot ra
    // B operands: feature f = 2 kq + s of the three weight matrices for column 16 t + i (zero beyond f_in)
    float bw[2][T], bi[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int f = 2 * kq + s;
        const bool fok = f < p.f_in;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float* w = t < 4 ? p.w_low : p.w_high;
            bw[s][t] = fok ? w[(long)f * p.ld_w + 16 * (t & 3) + i] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) bi[s][t] = fok ? p.w_mlp[(long)f * p.ld_w + 16 * t + i] : 0.f;
    }
    float mixm[K * K];
#pragma unroll
    for (int q = 0; q < K * K; ++q) mixm[q] = p.att_mix[q];
    const AcmDropCtx dc = acm_drop_ctx(p.post_drop);
    const bool unit = csr.vals == nullptr;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int n_quads = (csr.n_items + 3) >> 2;

    // One wave per quad, workgroups dispatched in list order: the pieces of the long rows first, then the rows by
    // decreasing length on a degree-sorted graph -- the hardware dispatcher does longest-first list scheduling.  (Persistent
    // waves with a static stride left those that drew the 1 300-neighbour pieces with 1.8x the mean load; a shared
    // atomic work counter serialised 42 k fetches on one L2 line: 667 -> 786 us.)
    {
        const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (q >= n_quads) return;
        const int wa = 4 * q + ga, wd = 4 * q + kq;
        const bool valid_a = wa < csr.n_items, valid_d = wd < csr.n_items;
        const AcmItem ia = csr.items[valid_a ? wa : 0], id = csr.items[valid_d ? wd : 0];
        const int beg_a = ia.begin, end_a = valid_a ? ia.end : ia.begin;
        const int len = end_a - beg_a;
        const int maxlen = max(max(__builtin_amdgcn_readlane(len, 0), __builtin_amdgcn_readlane(len, 4)),
                               max(__builtin_amdgcn_readlane(len, 8), __builtin_amdgcn_readlane(len, 12)));
        const int nb = (maxlen + 3) >> 2;        // wave-uniform
        float acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = 0.f;
        // software pipeline: column ids FOUR batches ahead, gathered rows TWO batches ahead of the MFMAs (a wave that
        // holds pieces of a hub row walks ~330 batches back to back: every exposed latency is on its critical path)
        const int pos0 = beg_a + ra;
        // branch-free loads (clamped addresses): guarded loads become exec-mask branches, and the compiler then waits
        // for EVERY outstanding load (s_waitcnt vmcnt(0)) before the MFMAs instead of only for the oldest
        auto id_at = [&](int b, bool& ok, float& av) {
            const int pos = pos0 + 4 * b;
            ok = pos < end_a;
            const int pc = ok ? pos : 0;
            av = unit ? 1.f : csr.vals[pc];
            return csr.indices[pc];
        };
        auto row_at = [&](int j, bool ok) {
            return *reinterpret_cast<const float2*>(p.xg + (long)(ok ? j : 0) * p.ld_xg + 2 * kq);
        };
        bool ok0, ok1, ok2, ok3;
        float a0v, a1v, a2v, a3v;
        int j0 = id_at(0, ok0, a0v), j1 = id_at(1, ok1, a1v), j2 = id_at(2, ok2, a2v), j3 = id_at(3, ok3, a3v);
        float2 x0 = row_at(j0, ok0), x1 = row_at(j1, ok1);
        (void)j0;
        float sx0 = 0.f, sx1 = 0.f;                        // sum over this lane's neighbours of its two A-operand features
        for (int b = 0; b < nb; ++b) {
            bool ok4;
            float a4v;
            const int j4 = id_at(b + 4, ok4, a4v);
            const float2 x2 = row_at(j2, ok2);
            const float e0 = ok0 ? a0v * x0.x : 0.f, e1 = ok0 ? a0v * x0.y : 0.f;     // idle slots contribute relu(0) = 0
            sx0 += e0;
            sx1 += e1;
            f32x4 d[T];
#pragma unroll
            for (int t = 0; t < T; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(e0, bw[0][t], zero4, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(e1, bw[1][t], d[t], 0, 0, 0);
            // relu(z) = (z + |z|) / 2: only sum |z| per edge (|.| is a free source modifier of v_add_f32: 4 VALU
            // instructions per tile instead of 8); sum z is linear in the gathered rows and comes out of one more MFMA
            // pass per quad below (from the lane's running sums sx0 / sx1 of its A operands)
#pragma unroll
            for (int t = 0; t < T; ++t)
                acc[t] += abs_add(d[t][0], d[t][1]) + abs_add(d[t][2], d[t][3]);
            ok0 = ok1, a0v = a1v, x0 = x1;
            ok1 = ok2, a1v = a2v, x1 = x2;
            ok2 = ok3, a2v = a3v, j2 = j3;
            ok3 = ok4, a3v = a4v, j3 = j4;
        }
        // sum_j relu(z_j) = (sum_j z_j + sum_j |z_j|) / 2, with sum_j z_j = (sum_j a_ij x_j) [W_L | W_H] on the matrix pipe: the
        // four A rows of item g carry the partial sums of its four neighbour slots
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(sx0, bw[0][t], zero4, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(sx1, bw[1][t], d, 0, 0, 0);
            acc[t] = 0.5f * (((d[0] + d[1]) + (d[2] + d[3])) + acc[t]);
        }
        // the rows' own projected features relu(x_i [W_L | W_H | W_I]) (what K1 writes in the literal path): A row 4 g of
        // the operand carries item g's own input row, so register 0 of lane group g holds its result
        float zs[12];
        {
            const float2 xi = (valid_a && ra == 0) ? *reinterpret_cast<const float2*>(p.xs + (long)ia.row * p.ld_xs + 2 * kq)
                                                   : make_float2(0.f, 0.f);
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const float b0 = t < T ? bw[0][t] : bi[0][t - T], b1 = t < T ? bw[1][t] : bi[1][t - T];
                f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(xi.x, b0, zero4, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(xi.y, b1, d, 0, 0, 0);
                zs[t] = relu1(d[0]);
            }
        }
        // ---- per 16-lane group: its item (row, slot)
        const int row = id.row, slot = id.slot;
        const long rr = valid_d ? row : 0;
        bool owner = valid_d && slot < 0;
        if (valid_d && slot >= 0) owner = csr.long_rows[csr.long_index[row]].slot_begin == slot;   // first piece of a long row
        if (owner) {
#pragma unroll
            for (int t = 0; t < T; ++t) p.zlh[rr * p.ld_zlh + 16 * t + i] = zs[t];
#pragma unroll
            for (int t = 0; t < 4; ++t) p.zi[rr * p.ld_zi + 16 * t + i] = zs[T + t];
        }
        if (valid_d && slot >= 0) {                        // a piece of a long row: raw sums to its slot
            float* ps = partial + (long)slot * (2 * F);
#pragma unroll
            for (int t = 0; t < T; ++t) ps[16 * t + i] = acc[t];
        }
        const bool full = valid_d && slot < 0;
        // ---- epilogue in the 16-lane x 4-column layout (lane i owns columns i, i + 16, i + 32, i + 48 of its group's row)
        const float rs = p.row_scale ? p.row_scale[rr] : 1.f;
        float H[K][4], pre[3][4];
        const float dg = K == 4 ? p.deg[rr] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            pre[0][t] = rs * acc[t];
            pre[1][t] = zs[4 + t] - rs * acc[4 + t];
            H[0][t] = pre[0][t];                           // ACMII: no ReLU after the filter
            H[1][t] = pre[1][t];
            H[2][t] = zs[T + t];
            if (K == 4) {                                  // structure channel: relu(A S) = relu(deg (A_low S) - S), ps = A_low S
                pre[2][t] = dg * p.ps[rr * p.ld_ps + i + 16 * t] - p.ss[rr * p.ld_ss + i + 16 * t];
                H[K - 1][t] = fmaxf(pre[2][t], 0.f);
            }
        }
        RowHead<K> rh;
        row_head<K>(hlds, mixm, acm_opaque(i), F, p.layernorm != 0, H, rh);
        float df[4];
        acm_drop4(dc, rr, i, df);
        if (full) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int col = i + 16 * t;
                float o = rh.alpha[0] * H[0][t] + rh.alpha[1] * H[1][t] + rh.alpha[2] * H[2][t];
                if (K == 4) o = fmaf(rh.alpha[K - 1], H[K - 1][t], o);
                o *= p.scale;
                if (p.post_relu) o = fmaxf(o, 0.f);
                if (p.post_scale) o *= p.post_scale[rr * p.ld_post_scale + col];
                if (p.post_drop.p > 0.f) o *= df[t];
                p.out[rr * p.ld_out + col] = o;
                p.pre[rr * p.ld_pre + col] = pre[0][t];
                p.pre[rr * p.ld_pre + F + col] = pre[1][t];
                if (K == 4) p.pre[rr * p.ld_pre + 2 * F + col] = pre[2][t];
            }
            if (i == 0)
                *reinterpret_cast<float4*>(p.att + rr * 4) =
                    make_float4(rh.alpha[0], rh.alpha[1], rh.alpha[2], K == 4 ? rh.alpha[K - 1] : 0.f);
        }
    }
}

// Long rows: one wave per row adds the pieces' partial sums in slot order and runs the same epilogue
// (lane l owns column l of each channel; 64-lane reductions).
template <int K>
__global__ __launch_bounds__(256) void acmii_fixup_kernel(acm_conv_acmii_fwd_t p, CsrView csr, const float* __restrict__ partial) {
    const int F = 64, lane = threadIdx.x & 63;
    const int w = acm_uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= csr.n_long) return;
    const AcmLongRow lr = csr.long_rows[w];
    const int row = acm_uniform(lr.row);
    float pl = 0.f, ph = 0.f;
    for (int s = lr.slot_begin; s < lr.slot_end; ++s) {
        pl += partial[(long)s * (2 * F) + lane];
        ph += partial[(long)s * (2 * F) + F + lane];
    }
    const float rs = p.row_scale ? p.row_scale[row] : 1.f;
    const float pre0 = rs * pl, pre1 = p.zlh[(long)row * p.ld_zlh + F + lane] - rs * ph;
    const float pre2 = K == 4 ? p.deg[row] * p.ps[(long)row * p.ld_ps + lane] - p.ss[(long)row * p.ld_ss + lane] : 0.f;
    float H[4][1] = {{pre0}, {pre1}, {p.zi[(long)row * p.ld_zi + lane]}, {fmaxf(pre2, 0.f)}}, hn[4][1], xhat[4][1];
    HeadOut ho;
    HeadParams hp;
#pragma unroll
    for (int c = 0; c < 4; ++c) hp.att_vec[c] = p.att_vec[c], hp.ln_w[c] = p.ln_weight[c], hp.ln_b[c] = p.ln_bias[c];
    hp.att_mix = p.att_mix;
    const LayWide<1> lay{lane};
    acm_head<LayWide<1>, K>(lay, F, p.layernorm, hp, H, hn, xhat, ho);
    float o = ho.alpha[0] * H[0][0] + ho.alpha[1] * H[1][0] + ho.alpha[2] * H[2][0];
    if (K == 4) o += ho.alpha[3] * H[3][0];
    o *= p.scale;
    if (p.post_relu) o = fmaxf(o, 0.f);
    if (p.post_scale) o *= p.post_scale[(long)row * p.ld_post_scale + lane];
    if (p.post_drop.p > 0.f) o *= acm_drop1(acm_drop_ctx(p.post_drop), row, lane);
    p.out[(long)row * p.ld_out + lane] = o;
    p.pre[(long)row * p.ld_pre + lane] = pre0;
    p.pre[(long)row * p.ld_pre + F + lane] = pre1;
    if (K == 4) p.pre[(long)row * p.ld_pre + 2 * F + lane] = pre2;
    if (lane == 0)
        *reinterpret_cast<float4*>(p.att + (long)row * 4) = make_float4(ho.alpha[0], ho.alpha[1], ho.alpha[2], K == 4 ? ho.alpha[3] : 0.f);
}

}  // namespace

// The long rows' second launch on its own: acm_conv_acmii_v_fwd (acm_conv_acmii_v.hip) leaves the pieces' raw sums in the same
// slots and the rows' own relu(x_i W_H) in zlh's second half -- everything this kernel reads.
int acm_acmii_fixup_launch(const acm_csr_t* a, const acm_conv_acmii_fwd_t* p, const float* partial, hipStream_t s) {
    if (!a->n_long) return ACM_OK;
    const CsrView cv = acm_view(a);
    const dim3 fg((unsigned)((a->n_long + 3) / 4));
    if (p->n_channels == 3) hipLaunchKernelGGL(acmii_fixup_kernel<3>, fg, dim3(256), 0, s, *p, cv, partial);
    else hipLaunchKernelGGL(acmii_fixup_kernel<4>, fg, dim3(256), 0, s, *p, cv, partial);
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}

// partial slots of the long rows' pieces ([n_slots, 2 F])
extern "C" int acm_conv_acmii_fwd_workspace_bytes(const acm_csr_t* a, size_t* bytes) {
    ACM_REQUIRE(a && bytes, ACM_EINVAL, "acm_conv_acmii_fwd_workspace_bytes: NULL argument");
    *bytes = (size_t)a->n_slots * 128 * sizeof(float) + 64;      // never zero: the caller always has a buffer to pass
    return ACM_OK;
}

extern "C" int acm_conv_acmii_fwd(const acm_csr_t* a, const acm_conv_acmii_fwd_t* p, void* workspace, size_t workspace_bytes,
                                  acm_stream_t stream) {
    ACM_REQUIRE(a && p, ACM_EINVAL, "acm_conv_acmii_fwd: NULL argument");
    ACM_REQUIRE(p->f_out == 64 && p->f_in >= 1 && p->f_in <= 8 && p->f_pad == 8, ACM_EUNSUPPORTED,
                "acm_conv_acmii_fwd: f_in %d f_pad %d f_out %d (needs f_in <= 8 = f_pad, f_out = 64)", p->f_in, p->f_pad,
                p->f_out);
    ACM_REQUIRE(p->xg && p->xs && p->w_low && p->w_high && p->w_mlp && p->att_mix && p->out && p->pre && p->att && p->zlh &&
                    p->zi, ACM_EINVAL, "acm_conv_acmii_fwd: NULL tensor pointer");
    ACM_REQUIRE(((uintptr_t)p->xg) % 8 == 0 && p->ld_xg % 2 == 0 && p->ld_xg >= p->f_pad && ((uintptr_t)p->xs) % 8 == 0 &&
                    p->ld_xs % 2 == 0 && p->ld_xs >= p->f_pad && ((uintptr_t)p->att) % 16 == 0, ACM_EINVAL,
                "acm_conv_acmii_fwd: xg / xs rows must be 8-byte aligned and f_pad long, att 16-byte aligned");
    const int K = p->n_channels;
    ACM_REQUIRE(K == 3 || K == 4, ACM_ESHAPE, "acm_conv_acmii_fwd: n_channels %d", K);
    ACM_REQUIRE(p->ld_w >= 64 && p->ld_out >= 64 && p->ld_pre >= 64 * (K - 1) && p->ld_zlh >= 128 && p->ld_zi >= 64, ACM_ESHAPE,
                "acm_conv_acmii_fwd: leading dimension too small");
    ACM_REQUIRE(K == 3 || (p->ps && p->ss && p->deg && p->ld_ps >= 64 && p->ld_ss >= 64), ACM_EINVAL,
                "acm_conv_acmii_fwd: structure-channel pointers NULL / leading dimensions too small");
    for (int c = 0; c < K; ++c) {
        ACM_REQUIRE(p->att_vec[c], ACM_EINVAL, "acm_conv_acmii_fwd: att_vec[%d] NULL", c);
        ACM_REQUIRE(!p->layernorm || (p->ln_weight[c] && p->ln_bias[c]), ACM_EINVAL, "acm_conv_acmii_fwd: LayerNorm pointers NULL");
    }
    size_t need = 0;
    acm_conv_acmii_fwd_workspace_bytes(a, &need);
    ACM_REQUIRE(workspace && workspace_bytes >= need, ACM_ENOMEM,
                "acm_conv_acmii_fwd: workspace %zu B < required %zu B", workspace_bytes, need);
    ACM_REQUIRE(a->n_long == 0 || a->long_index, ACM_EUNSUPPORTED, "acm_conv_acmii_fwd: handle without a long-row index");
    if (a->n_rows == 0 || a->n_items == 0) return ACM_OK;
    hipStream_t s = (hipStream_t)stream;
    const CsrView cv = acm_view(a);
    const int grid = (int)((a->n_items + 15) / 16);      // a wave takes four items, a workgroup sixteen
    if (K == 3) hipLaunchKernelGGL(acmii_fwd_kernel<3>, dim3(grid), dim3(256), 0, s, *p, cv, (float*)workspace);
    else hipLaunchKernelGGL(acmii_fwd_kernel<4>, dim3(grid), dim3(256), 0, s, *p, cv, (float*)workspace);
    ACM_CHECK_HIP(hipGetLastError());
    if (a->n_long) {
        const dim3 fg((unsigned)((a->n_long + 3) / 4));
        if (K == 3) hipLaunchKernelGGL(acmii_fixup_kernel<3>, fg, dim3(256), 0, s, *p, cv, (const float*)workspace);
        else hipLaunchKernelGGL(acmii_fixup_kernel<4>, fg, dim3(256), 0, s, *p, cv, (const float*)workspace);
        ACM_CHECK_HIP(hipGetLastError());
    }
    return ACM_OK;
}
