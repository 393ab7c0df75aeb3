// Row-local stages of the aggregate-first ACM layer in the TRANSPOSED matrix-core layout (gfx950), for the
// reference's hidden width: three channels, f_pad = 8, F = 64 (ACM-Geometric/layers.py:57-63,101-108 after
// P = A_low X has been gathered).
//
// The older row-local kernels (acm_conv_agg.hip) give a matrix row to a 16-lane group: the projections
// P W_L, (X - P) W_H, X W_I are 96 FMAs per lane and row, every reduction of the head is a 16-lane DPP tree per row,
// and each per-row scalar (sigmoid, softmax) is computed by all 16 lanes.  That made them VALU-bound at 5x their
// stream time.  Here a wave takes SIXTEEN rows per step and lets v_mfma_f32_16x16x4_f32 produce the projections
// transposed:
//
//     D^T[col][row] = sum_f W_c[f][col] * A_c[row][f]      A operand = weights (24 loop-invariant registers per lane),
//                                                           B operand = P / X - P / X of row `m` (one dword per lane)
//
// so lane (g, m) ends up with row m's columns 16 t + 4 g + r (t, r = 0..3): a WHOLE row sits in the four lanes
// m, m + 16, m + 32, m + 48.  Consequences: no VALU work for the projections (the matrix pipe runs beside the VALU
// of the other waves), a head reduction is 15 in-lane adds + one cross-row sum (v_permlane16/32_swap) for 16 rows at
// once instead of a DPP tree per row, the per-row scalars are computed 4x redundantly instead of 16x, and the output
// row is stored as four 16-byte pieces per lane.  The MFMA is an exact k-ordered fmaf chain, so the projections are
// bit-identical to the older kernels'; the head statistics differ by summation order only.
#include "acm_conv_device.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ACM_E16_NEXT_LDS (64 * 8)

// sum over the four lanes that hold one row (lanes m, m + 16, m + 32, m + 48); result in all four
__device__ __forceinline__ float row4_sum(float v) { return acm_cross_row_sum(v); }

template <bool LN, bool NEXT>
__device__ __forceinline__ void epi16_body(const acm_conv_agg_fwd_t& p, int n_rows) {
    __shared__ __attribute__((aligned(16))) float ulds[3 * 64 + (NEXT ? ACM_E16_NEXT_LDS : 0)];
    const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    // u_c = gamma_c (.) att_vec_c (LayerNorm folded into the attention vector): with d = H - mean,
    //   s_c = sum_col (d * rstd * gamma + beta) * v = rstd * sum_col d * u_c + c0_c,   c0_c = sum_col beta_c * v_c
    for (int idx = threadIdx.x; idx < 192; idx += 256) {
        const int c = idx >> 6, col = idx & 63;
        const float* av = c == 0 ? p.att_vec[0] : (c == 1 ? p.att_vec[1] : p.att_vec[2]);
        float u = av[col];
        if (LN) {
            const float* gw = c == 0 ? p.ln_weight[0] : (c == 1 ? p.ln_weight[1] : p.ln_weight[2]);
            u *= gw[col];
        }
        ulds[idx] = u;
    }
    if (NEXT) {                     // [col][8] = [W_L'(col, :) | W_H'(col, :) | W_I'(col, :) | 0]
        float* nlds = ulds + 192;
        for (int idx = threadIdx.x; idx < ACM_E16_NEXT_LDS; idx += 256) {
            const int col = idx >> 3, j = idx & 7, c = j / p.next_f, q = j % p.next_f;
            const float* w = c == 0 ? p.next_w_low : (c == 1 ? p.next_w_high : p.next_w_mlp);
            nlds[idx] = (c < 3) ? w[(long)col * p.next_ld_w + q] : 0.f;
        }
    }
    float c0[3] = {0.f, 0.f, 0.f};
    if (LN) {
#pragma unroll
        for (int c = 0; c < 3; ++c) c0[c] = acm_group_sum<64>(p.ln_bias[c][lane] * p.att_vec[c][lane]);
    }
    // A operands: W_c[f = 4 kb + g][col = 16 t + m]
    float wreg[3][2][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* w = c == 0 ? p.w_low : (c == 1 ? p.w_high : p.w_mlp);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                wreg[c][kb][t] = (4 * kb + g < p.f_in) ? w[(long)(4 * kb + g) * p.ld_w + 16 * t + m] : 0.f;
    }
    float mixm[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) mixm[q] = p.att_mix[q];
    __syncthreads();
    const AcmDropCtx dc = acm_drop_ctx(p.post_drop);
    const float lo_a = p.relu_after ? 0.f : -INFINITY, lo_m = p.relu_mlp ? 0.f : -INFINITY;
    const float lo_post = p.post_relu ? 0.f : -INFINITY;
    const unsigned ld_agg = (unsigned)p.ld_agg, ld_xs = (unsigned)p.ld_xs, ld_out = (unsigned)p.ld_out;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;

    int base = wave * 16;
    if (base >= n_rows) return;
    // the operands of the NEXT step are requested before this step's math (one step of loads in flight)
    float nPa, nPb, nxa, nxb;
    {
        const unsigned rr = (unsigned)min(base + m, n_rows - 1);
        nPa = p.agg[rr * ld_agg + g], nPb = p.agg[rr * ld_agg + 4 + g];
        nxa = p.xs[rr * ld_xs + g], nxb = p.xs[rr * ld_xs + 4 + g];
    }
    for (; base < n_rows; base += nwaves * 16) {
        const int row = base + m;
        const bool valid = row < n_rows;
        const unsigned rr = (unsigned)(valid ? row : n_rows - 1);
        const float Pa = nPa, Pb = nPb, xa = nxa, xb = nxb;
        {
            const int nb = base + nwaves * 16;
            const unsigned r2 = (unsigned)min(nb + m, n_rows - 1);
            nPa = p.agg[r2 * ld_agg + g], nPb = p.agg[r2 * ld_agg + 4 + g];
            nxa = p.xs[r2 * ld_xs + g], nxb = p.xs[r2 * ld_xs + 4 + g];
        }
        if (p.agg_copy && valid) {                 // the backward's operands (input pipeline): the rows just read
            p.agg_copy[rr * (unsigned)p.ld_agg_copy + g] = Pa;
            p.agg_copy[rr * (unsigned)p.ld_agg_copy + 4 + g] = Pb;
            p.xs_copy[rr * (unsigned)p.ld_xs_copy + g] = xa;
            p.xs_copy[rr * (unsigned)p.ld_xs_copy + 4 + g] = xb;
        }
        // (an opaque copy of the lane's group index: the LDS operands below depend on the lane only, and hoisted out of the
        //  row loop they would pin 48 .. 144 registers)
        const int gq = acm_opaque(g);
        const float opa[3] = {Pa, xa - Pa, xa}, opb[3] = {Pb, xb - Pb, xb};
        f32x4 D[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                D[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[c][0][t], opa[c], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                D[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[c][1][t], opb[c], D[c][t], 0, 0, 0);
        // ---- head: statistics and attention scalars of row m (four lanes per row)
        float mean[3], rstd[3], gs[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float lo = c < 2 ? lo_a : lo_m;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) D[c][t][r] = fmaxf(D[c][t][r], lo);
            float dot = 0.f;
            if (LN) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) s += (D[c][t][0] + D[c][t][1]) + (D[c][t][2] + D[c][t][3]);
                const float mu = row4_sum(s) * (1.0f / 64.0f);
                float q = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(ulds + c * 64 + 16 * t + 4 * gq);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float d = D[c][t][r] - mu;
                        q = fmaf(d, d, q);
                        dot = fmaf(d, u[r], dot);
                    }
                }
                mean[c] = mu;
                rstd[c] = acm_rsqrt(row4_sum(q) * (1.0f / 64.0f) + ACM_LN_EPS);
                dot = fmaf(rstd[c], row4_sum(dot), c0[c]);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(ulds + c * 64 + 16 * t + 4 * gq);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dot = fmaf(D[c][t][r], u[r], dot);
                }
                mean[c] = 0.f;
                rstd[c] = 1.f;
                dot = row4_sum(dot);
            }
            gs[c] = acm_rcp(1.0f + acm_exp(-dot));
        }
        float al[3];
        {
            float lg[3], mx = -INFINITY, den = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) a = fmaf(gs[c], mixm[c * 3 + j], a);
                lg[j] = a * (1.0f / 3.0f);
                mx = fmaxf(mx, lg[j]);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                lg[j] = acm_exp(lg[j] - mx);
                den += lg[j];
            }
            const float inv = acm_rcp(den);
#pragma unroll
            for (int j = 0; j < 3; ++j) al[j] = lg[j] * inv;
        }
        if (valid) {
            if (p.head_stats && g == 0) {
                float* hs = p.head_stats + rr * (unsigned)p.ld_head_stats;
                reinterpret_cast<float4*>(hs)[0] = make_float4(mean[0], mean[1], mean[2], rstd[0]);
                reinterpret_cast<float4*>(hs)[1] = make_float4(rstd[1], rstd[2], gs[0], gs[1]);
                reinterpret_cast<float4*>(hs)[2] = make_float4(gs[2], al[0], al[1], al[2]);
            }
            if (g == 1) *reinterpret_cast<float4*>(p.att + (size_t)rr * 4) = make_float4(al[0], al[1], al[2], 0.f);
        }
        // ---- mix, post-op, store; the row's next-layer projection
        const float a0 = al[0] * p.scale, a1 = al[1] * p.scale, a2 = al[2] * p.scale;
        float z8[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x4 o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[t][r] = fmaxf(fmaf(a2, D[2][t][r], fmaf(a1, D[1][t][r], a0 * D[0][t][r])), lo_post);
        if (p.post_scale) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 s = *reinterpret_cast<const f32x4*>(p.post_scale + rr * (unsigned)p.ld_post_scale + 16 * t + 4 * g);
                o[t] *= s;
            }
        }
        if (dc.on) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned w[4];
                acm_philox7(dc, row, 4 * g + r, w);      // word t <-> column 16 t + (4 g + r): the mask of acm_drop4
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t][r] *= (w[t] >= dc.thresh) ? dc.inv_keep : 0.f;
            }
        }
        if (valid) {
#pragma unroll
            for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(p.out + rr * ld_out + 16 * t + 4 * g) = o[t];
        }
        if (NEXT) {
            const float* nlds = ulds + 192;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 wa = *reinterpret_cast<const float4*>(nlds + (16 * t + 4 * gq + r) * 8);
                    const float2 wb = *reinterpret_cast<const float2*>(nlds + (16 * t + 4 * gq + r) * 8 + 4);
                    z8[0] = fmaf(o[t][r], wa.x, z8[0]); z8[1] = fmaf(o[t][r], wa.y, z8[1]);
                    z8[2] = fmaf(o[t][r], wa.z, z8[2]); z8[3] = fmaf(o[t][r], wa.w, z8[3]);
                    z8[4] = fmaf(o[t][r], wb.x, z8[4]); z8[5] = fmaf(o[t][r], wb.y, z8[5]);
                }
            const int nf = p.next_f;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                z8[j] = row4_sum(z8[j]);
                if (p.next_relu) z8[j] = fmaxf(z8[j], 0.f);
            }
            if (valid && g == 2) {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    if (j < 2 * nf) p.next_zlh[(size_t)rr * p.ld_next_zlh + j] = z8[j];
                    else if (j < 3 * nf) p.next_zi[(size_t)rr * p.ld_next_zi + (j - 2 * nf)] = z8[j];
                }
            }
        }
    }
}

template <bool LN, bool NEXT>
__global__ __launch_bounds__(256) void agg_epi16_kernel(acm_conv_agg_fwd_t p, int n_rows) {
    epi16_body<LN, NEXT>(p, n_rows);
}
// the same capped to the registers of four waves per SIMD (a handful of spilled registers with LayerNorm)
template <bool LN, bool NEXT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void agg_epi16_cap4_kernel(acm_conv_agg_fwd_t p, int n_rows) {
    epi16_body<LN, NEXT>(p, n_rows);
}

}  // namespace

// The row-local forward stage over an existing P = A_low X (p->agg).  Returns ACM_OK after a launch, -1 when the
// configuration is not the one this kernel is written for (the caller then runs agg_epilogue_kernel), or an error.
int acm_agg_epi16(const acm_conv_agg_fwd_t* p, int64_t n_rows, bool* next_done, hipStream_t s) {
    *next_done = false;
    if (p->n_channels != 3 || p->f_pad != 8 || p->f_out != 64 || getenv("ACM_EPI16_OFF") != nullptr) return -1;
    int64_t ld_max = 64;
    for (int64_t ld : {p->ld_agg, p->ld_xs, p->ld_out, p->ld_head_stats, p->ld_post_scale, p->ld_agg_copy, p->ld_xs_copy})
        ld_max = ld > ld_max ? ld : ld_max;
    if (n_rows * ld_max >= (int64_t)INT32_MAX) return -1;                  // 32-bit element offsets
    if ((((uintptr_t)p->out) % 16) != 0 || (p->ld_out % 4) != 0) return -1;
    if (p->post_scale && ((((uintptr_t)p->post_scale) % 16) != 0 || (p->ld_post_scale % 4) != 0)) return -1;
    const bool next = p->next_f > 0 && getenv("ACM_AGG_NO_NEXT") == nullptr;
    int grid = (int)((n_rows + 63) / 64);
    int cap = 1024;                                  // four workgroups (sixteen waves) per CU
    if (const char* env = getenv("ACM_EPI16_BLOCKS")) {
        const int v = atoi(env);
        if (v >= 1) cap = v;
    }
    if (grid > cap) grid = cap;
    const bool cap4 = getenv("ACM_EPI16_CAP4") != nullptr;
#define ACM_E16(KERNEL)                                                                                              \
    do {                                                                                                             \
        if (p->layernorm) {                                                                                          \
            if (next) hipLaunchKernelGGL((KERNEL<true, true>), dim3(grid), dim3(256), 0, s, *p, (int)n_rows);        \
            else hipLaunchKernelGGL((KERNEL<true, false>), dim3(grid), dim3(256), 0, s, *p, (int)n_rows);            \
        } else {                                                                                                     \
            if (next) hipLaunchKernelGGL((KERNEL<false, true>), dim3(grid), dim3(256), 0, s, *p, (int)n_rows);       \
            else hipLaunchKernelGGL((KERNEL<false, false>), dim3(grid), dim3(256), 0, s, *p, (int)n_rows);           \
        }                                                                                                            \
    } while (0)
    if (cap4) ACM_E16(agg_epi16_cap4_kernel);
    else ACM_E16(agg_epi16_kernel);
#undef ACM_E16
    ACM_CHECK_HIP(hipGetLastError());
    *next_done = next;
    return ACM_OK;
}
