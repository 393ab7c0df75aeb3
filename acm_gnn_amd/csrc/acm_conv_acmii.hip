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
//   one wave per work item, batches of 16 neighbours = the M dimension of v_mfma_f32_16x16x4_f32;
//   A[i][k] = x_{j_i}[feature], lane (i = lane & 15, kq = lane >> 4) fetches the float2 (2 kq, 2 kq + 1) of its
//             neighbour's row (K-step s uses feature 2 kq + s: one 8-byte load per lane and batch);
//   B[k][n] = [W_L | W_H][2 kq + s][16 t + i], 16 loop-invariant registers per lane (waves are persistent);
//   D tile t = 16 neighbours x 16 columns; ReLU and the sum over the tile's four row registers on the VALU,
//   accumulated per lane, the four kq groups summed once per row (v_permlane16/32_swap).
//
// 2 x 8 x 128 FLOP per edge = 28 GFLOP per pass; at the 157 TFLOP/s fp32 MFMA peak 0.18 ms.
// Values of an explicit operator (a_ij > 0) scale the gathered row before the product: relu(a z) = a relu(z).
// The same kernel writes what the literal path's K1 GEMM would have written for the backward (the row's own
// relu(x_i [W_L | W_H | W_I]): K4's masks and self term, K3's s_mlp), so it replaces K1 + K2.
#include "acm_conv_device.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K>
struct RowOut {
    float H[K][4];
};

__global__ __launch_bounds__(256) void acmii_fwd_kernel(acm_conv_acmii_fwd_t p, CsrView csr, float* __restrict__ partial) {
    constexpr int K = 3, T = 8;                  // T = tiles of 16 gathered columns: [Z_L (4) | Z_H (4)], F = 64
    __shared__ __attribute__((aligned(16))) float hlds[3 * K * 64];
    const int F = 64;
    stage_head_params<K>(hlds, p.att_vec, p.ln_weight, p.ln_bias, p.layernorm, F);
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
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

    for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < csr.n_items; w += gridDim.x * 4) {
        const AcmItem it = csr.items[acm_uniform(w)];
        const int row = acm_uniform(it.row), begin = acm_uniform(it.begin), end = acm_uniform(it.end),
                  slot = acm_uniform(it.slot);
        float acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = 0.f;
        // software pipeline over batches of 16 neighbours: the next batch's ids and rows are requested before the
        // current batch's MFMAs
        int k0 = begin;
        bool v = k0 + i < end;
        int j = v ? csr.indices[k0 + i] : 0;
        float a = v ? (unit ? 1.f : csr.vals[k0 + i]) : 0.f;
        float2 x2 = v ? *reinterpret_cast<const float2*>(p.xg + (long)j * p.ld_xg + 2 * kq) : make_float2(0.f, 0.f);
        while (k0 < end) {
            const int k1 = k0 + 16;
            const bool nv = k1 + i < end;
            const int nj = nv ? csr.indices[k1 + i] : 0;
            const float na = nv ? (unit ? 1.f : csr.vals[k1 + i]) : 0.f;
            const float2 nx2 = nv ? *reinterpret_cast<const float2*>(p.xg + (long)nj * p.ld_xg + 2 * kq) : make_float2(0.f, 0.f);
            const float a0 = v ? a * x2.x : 0.f, a1 = v ? a * x2.y : 0.f;     // idle slots contribute relu(0) = 0
            f32x4 d[T];
#pragma unroll
            for (int t = 0; t < T; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bw[0][t], zero4, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bw[1][t], d[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t)
                acc[t] += (fmaxf(d[t][0], 0.f) + fmaxf(d[t][1], 0.f)) + (fmaxf(d[t][2], 0.f) + fmaxf(d[t][3], 0.f));
            k0 = k1;
            v = nv, j = nj, a = na, x2 = nx2;
        }
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = acm_cross_row_sum(acc[t]);   // every lane: column 16 t + i of [P_L | P_H]
        const bool owner = slot < 0 || (csr.long_index && csr.long_rows[csr.long_index[row]].slot_begin == slot);
        // the row's own projected features relu(x_i [W_L | W_H | W_I]) (what K1 writes in the literal path)
        float zs[12];
        if (owner) {
            const float2 xi = *reinterpret_cast<const float2*>(p.xs + (long)row * p.ld_xs + 2 * kq);
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const float b0 = t < T ? bw[0][t] : bi[0][t - T], b1 = t < T ? bw[1][t] : bi[1][t - T];
                zs[t] = fmaxf(acm_cross_row_sum(fmaf(xi.x, b0, xi.y * b1)), 0.f);
            }
            if (kq == 0) {
#pragma unroll
                for (int t = 0; t < T; ++t) p.zlh[(long)row * p.ld_zlh + 16 * t + i] = zs[t];
#pragma unroll
                for (int t = 0; t < 4; ++t) p.zi[(long)row * p.ld_zi + 16 * t + i] = zs[T + t];
            }
        }
        if (slot >= 0) {                                   // a piece of a long row: raw sums to its slot
            if (kq == 0) {
                float* ps = partial + (long)slot * (2 * F);
#pragma unroll
                for (int t = 0; t < T; ++t) ps[16 * t + i] = acc[t];
            }
            continue;
        }
        // ---- epilogue in the 16-lane x 4-column layout (LayGrouped<4>: lane i owns columns i, i + 16, i + 32, i + 48),
        // identical in the four groups; group 0 stores
        const float rs = p.row_scale ? p.row_scale[row] : 1.f;
        float H[K][4], pre[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            pre[0][t] = rs * acc[t];
            pre[1][t] = zs[4 + t] - rs * acc[4 + t];
            H[0][t] = pre[0][t];                           // ACMII: no ReLU after the filter
            H[1][t] = pre[1][t];
            H[2][t] = zs[T + t];
        }
        RowHead<K> rh;
        row_head<K>(hlds, mixm, acm_opaque(i), F, p.layernorm != 0, H, rh);
        float df[4];
        acm_drop4(dc, row, i, df);
        if (kq == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int col = i + 16 * t;
                float o = p.scale * (rh.alpha[0] * H[0][t] + rh.alpha[1] * H[1][t] + rh.alpha[2] * H[2][t]);
                if (p.post_relu) o = fmaxf(o, 0.f);
                if (p.post_scale) o *= p.post_scale[(long)row * p.ld_post_scale + col];
                if (p.post_drop.p > 0.f) o *= df[t];
                p.out[(long)row * p.ld_out + col] = o;
                p.pre[(long)row * p.ld_pre + col] = pre[0][t];
                p.pre[(long)row * p.ld_pre + F + col] = pre[1][t];
            }
            if (i == 0)
                *reinterpret_cast<float4*>(p.att + (long)row * 4) = make_float4(rh.alpha[0], rh.alpha[1], rh.alpha[2], 0.f);
        }
    }
}

// Long rows: one wave per row adds the pieces' partial sums in slot order and runs the same epilogue
// (lane l owns column l of each channel; 64-lane reductions).
__global__ __launch_bounds__(256) void acmii_fixup_kernel(acm_conv_acmii_fwd_t p, CsrView csr, const float* __restrict__ partial) {
    constexpr int K = 3;
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
    float H[4][1] = {{pre0}, {pre1}, {p.zi[(long)row * p.ld_zi + lane]}, {0.f}}, hn[4][1], xhat[4][1];
    HeadOut ho;
    HeadParams hp;
#pragma unroll
    for (int c = 0; c < 4; ++c) hp.att_vec[c] = p.att_vec[c], hp.ln_w[c] = p.ln_weight[c], hp.ln_b[c] = p.ln_bias[c];
    hp.att_mix = p.att_mix;
    const LayWide<1> lay{lane};
    acm_head<LayWide<1>, K>(lay, F, p.layernorm, hp, H, hn, xhat, ho);
    float o = p.scale * (ho.alpha[0] * H[0][0] + ho.alpha[1] * H[1][0] + ho.alpha[2] * H[2][0]);
    if (p.post_relu) o = fmaxf(o, 0.f);
    if (p.post_scale) o *= p.post_scale[(long)row * p.ld_post_scale + lane];
    if (p.post_drop.p > 0.f) o *= acm_drop1(acm_drop_ctx(p.post_drop), row, lane);
    p.out[(long)row * p.ld_out + lane] = o;
    p.pre[(long)row * p.ld_pre + lane] = pre0;
    p.pre[(long)row * p.ld_pre + F + lane] = pre1;
    if (lane == 0) *reinterpret_cast<float4*>(p.att + (long)row * 4) = make_float4(ho.alpha[0], ho.alpha[1], ho.alpha[2], 0.f);
}

}  // namespace

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
    ACM_REQUIRE(p->ld_w >= 64 && p->ld_out >= 64 && p->ld_pre >= 128 && p->ld_zlh >= 128 && p->ld_zi >= 64, ACM_ESHAPE,
                "acm_conv_acmii_fwd: leading dimension too small");
    for (int c = 0; c < 3; ++c) {
        ACM_REQUIRE(p->att_vec[c], ACM_EINVAL, "acm_conv_acmii_fwd: att_vec[%d] NULL", c);
        ACM_REQUIRE(!p->layernorm || (p->ln_weight[c] && p->ln_bias[c]), ACM_EINVAL, "acm_conv_acmii_fwd: LayerNorm pointers NULL");
    }
    const size_t need = (size_t)a->n_slots * 128 * sizeof(float);
    ACM_REQUIRE(workspace_bytes >= need && (need == 0 || workspace), ACM_ENOMEM,
                "acm_conv_acmii_fwd: workspace %zu B < required %zu B", workspace_bytes, need);
    ACM_REQUIRE(a->n_long == 0 || a->long_index, ACM_EUNSUPPORTED, "acm_conv_acmii_fwd: handle without a long-row index");
    if (a->n_rows == 0 || a->n_items == 0) return ACM_OK;
    hipStream_t s = (hipStream_t)stream;
    const CsrView cv = acm_view(a);
    int grid = (int)((a->n_items + 3) / 4);
    // persistent waves (the 24 weight registers are loaded once per wave): four workgroups per CU keep every SIMD's
    // matrix pipe fed while other waves sit in their epilogues
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(acmii_fwd_kernel, dim3(grid), dim3(256), 0, s, *p, cv, (float*)workspace);
    ACM_CHECK_HIP(hipGetLastError());
    if (a->n_long) {
        hipLaunchKernelGGL(acmii_fixup_kernel, dim3((unsigned)((a->n_long + 3) / 4)), dim3(256), 0, s, *p, cv,
                           (const float*)workspace);
        ACM_CHECK_HIP(hipGetLastError());
    }
    return ACM_OK;
}
