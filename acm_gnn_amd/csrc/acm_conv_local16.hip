// K3 of the literal ACM layer at the reference's hidden width (three channels, F = 64) with SIXTEEN rows per wave
// (ACM-Geometric/layers.py:57-63,101-108 backwards: the autograd replay of attention3 + LayerNorm + the mix, train.py:135).
//
// The older kernel (acm_conv.hip: conv_bwd_local_grouped_kernel) gives a row to a 16-lane group, columns m + 16 i: every load
// and store is a dword per lane, every reduction of the head a 16-lane DPP tree per row, every per-row scalar computed 16
// times.  It moves 1.8 KB per row (pre_L, pre_H, Z_I, grad_out in; G_L, G_H, G_I out) at 3.0 TB/s.  Here lane (g, m) holds
// columns 16 t + 4 g + r of row m (the layout of acm_conv_agg16.hip): 16-byte loads and stores, head sums = 15 in-lane adds +
// one cross-row exchange for 16 rows at once, per-row scalars 4x redundant, the head-parameter sums by reduce-scatter.
// Same partial-vector layout as the older kernels, so the same second phase (bwd_local_reduce) finishes the job.
#include "acm_rows16_device.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LN>
__global__ __launch_bounds__(256) void bwd_local16_kernel(acm_conv_bwd_local_t p, int n_rows, float* __restrict__ partial) {
    constexpr int NPG = 3 * 3 * 64 + 9;
    __shared__ __attribute__((aligned(16))) float lds[4 * NPG];
    const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    float* hl = lds;                       // [att_vec | gamma | beta][c][col]: 576 floats
    float* ul = lds + 576;                 // u_c = att_vec_c (.) gamma_c: 192 floats   (both dead before the slabs alias them)
    for (int idx = threadIdx.x; idx < 576; idx += 256) {
        const int arr = idx / 192, c = (idx / 64) % 3, col = idx & 63;
        float v;
        if (arr == 0) v = p.att_vec[c][col];
        else if (LN) v = arr == 1 ? p.ln_weight[c][col] : p.ln_bias[c][col];
        else v = arr == 1 ? 1.f : 0.f;
        hl[idx] = v;
    }
    if (threadIdx.x < 192) {
        const int c = threadIdx.x >> 6, col = threadIdx.x & 63;
        float u = p.att_vec[c][col];
        if (LN) u *= p.ln_weight[c][col];
        ul[threadIdx.x] = u;
    }
    float c1[3], c0[3];                    // mean_col(u_c); sum_col beta_c v_c
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float u = p.att_vec[c][lane];
        c0[c] = LN ? acm_group_sum<64>(p.ln_bias[c][lane] * u) : 0.f;
        if (LN) u *= p.ln_weight[c][lane];
        c1[c] = acm_group_sum<64>(u) * (1.0f / 64.0f);
    }
    float mixm[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) mixm[q] = p.att_mix[q];
    __syncthreads();
    const float lo_a = p.relu_after ? 0.f : -INFINITY, lo_m = p.relu_mlp ? 0.f : -INFINITY;
    const AcmDropCtx dc = acm_drop_ctx(p.post_drop);
    const unsigned ld_pre = (unsigned)p.ld_pre, ld_zi = (unsigned)p.ld_s_mlp, ld_go = (unsigned)p.ld_grad_out,
                   ld_gl = (unsigned)p.ld_g_low, ld_gh = (unsigned)p.ld_g_high, ld_gm = (unsigned)p.ld_g_mlp;
    const int wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    const float wq = g == 0 ? 1.f : 0.f;   // a row's scalars sit in four lanes: one of them accumulates
    float pA[3], pS[3], dmix[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) pS[c] = pA[c] = 0.f;
#pragma unroll
    for (int q = 0; q < 9; ++q) dmix[q] = 0.f;
    for (int base = wave * 16; base < n_rows; base += nwaves * 16) {
        const int row = base + m;
        const bool valid = row < n_rows;
        const unsigned r1 = (unsigned)min(row, n_rows - 1);
        const int gq = acm_opaque(g), mq = acm_opaque(m);
        f32x4 D[3][4], dO[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            D[0][t] = *reinterpret_cast<const f32x4*>(p.pre + r1 * ld_pre + 16 * t + 4 * g);
            D[1][t] = *reinterpret_cast<const f32x4*>(p.pre + r1 * ld_pre + 64 + 16 * t + 4 * g);
            D[2][t] = *reinterpret_cast<const f32x4*>(p.s_mlp + r1 * ld_zi + 16 * t + 4 * g);
            dO[t] = *reinterpret_cast<const f32x4*>(p.grad_out + r1 * ld_go + 16 * t + 4 * g);
        }
        // ---- the head again: statistics and attention scalars of row m
        float mean[3], rstd[3], gsig[3], al[3];
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
                    const f32x4 u = *reinterpret_cast<const f32x4*>(ul + c * 64 + 16 * t + 4 * gq);
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
                    const f32x4 u = *reinterpret_cast<const f32x4*>(ul + c * 64 + 16 * t + 4 * gq);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dot = fmaf(D[c][t][r], u[r], dot);
                }
                mean[c] = 0.f;
                rstd[c] = 1.f;
                dot = row4_sum(dot);
            }
            gsig[c] = acm_rcp(1.0f + acm_exp(-dot));
        }
        {
            float lg[3], mx = -INFINITY, den = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) a = fmaf(gsig[c], mixm[c * 3 + j], a);
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
        // ---- undo the forward's fused post-op on the incoming gradient: ReLU of the mixed row (recomputed), dropout (regenerated)
        if (p.post_relu) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float raw = fmaf(al[2], D[2][t][r], fmaf(al[1], D[1][t][r], al[0] * D[0][t][r]));
                    dO[t][r] = raw * p.scale > 0.f ? dO[t][r] : 0.f;
                }
        }
        if (dc.on) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned w[4];
                acm_philox7(dc, row, 4 * g + r, w);      // word t <-> column 16 t + (4 g + r): the mask of acm_drop4
#pragma unroll
                for (int t = 0; t < 4; ++t) dO[t][r] = w[t] >= dc.thresh ? dO[t][r] * dc.inv_keep : 0.f;
            }
        }
        if (!valid) {
#pragma unroll
            for (int t = 0; t < 4; ++t) dO[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // ---- mix / softmax / sigmoid backward: ds_c = dL/ds_c per row
        float dal[3], ds[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float part = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fmaf(dO[t][r], D[c][t][r], part);
            dal[c] = p.scale * row4_sum(part);
        }
        {
            const float dot = fmaf(al[2], dal[2], fmaf(al[1], dal[1], al[0] * dal[0]));
            float dlg[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) dlg[j] = al[j] * (dal[j] - dot);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float dg = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    dg = fmaf(dlg[j], mixm[c * 3 + j], dg);
                    dmix[c * 3 + j] = fmaf(wq * gsig[c], dlg[j] * (1.0f / 3.0f), dmix[c * 3 + j]);
                }
                ds[c] = dg * (1.0f / 3.0f) * gsig[c] * (1.f - gsig[c]);
                pS[c] = fmaf(wq, ds[c], pS[c]);
            }
        }
        const float gsc = (valid && p.g_scale) ? p.g_scale[r1] : 1.f;
        // ---- one channel at a time: row sums for the parameter gradients, then G_c straight to memory
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float lo = c < 2 ? lo_a : lo_m;
            const float aal = p.scale * al[c];
            float contrib[16], t2 = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(ul + c * 64 + 16 * t + 4 * gq);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = LN ? (D[c][t][r] - mean[c]) * rstd[c] : D[c][t][r];
                    contrib[4 * t + r] = ds[c] * xh;
                    if (LN) t2 = fmaf(u[r], xh, t2);
                }
            }
            pA[c] += row_reduce_scatter16(contrib, mq);
            const float m1 = LN ? ds[c] * c1[c] : 0.f, m2 = LN ? ds[c] * row4_sum(t2) * (1.0f / 64.0f) : 0.f;
            float* dst = c == 0 ? p.g_low + r1 * ld_gl : (c == 1 ? p.g_high + r1 * ld_gh : p.g_mlp + r1 * ld_gm);
            const float sc = c < 2 ? gsc : 1.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(ul + c * 64 + 16 * t + 4 * gq);
                f32x4 G;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v;
                    if (LN) {
                        const float xh = (D[c][t][r] - mean[c]) * rstd[c];
                        v = fmaf(aal, dO[t][r], rstd[c] * (fmaf(ds[c], u[r], -m1) - xh * m2));
                    } else {
                        v = fmaf(aal, dO[t][r], ds[c] * u[r]);
                    }
                    G[r] = D[c][t][r] > lo ? sc * v : 0.f;
                }
                if (valid) *reinterpret_cast<f32x4*>(dst + 16 * t + 4 * g) = G;
            }
        }
    }
    // ---- head-parameter sums over the 16 row-lanes, then the block's partial vector in the older kernels' layout:
    // [d att_vec | d gamma | d beta][c][col] | d att_mix[c][j]
    // value i = 4 t + r of lane (g, m = i) is column 16 t + 4 g + r: one column of A_c per lane
    const int mycol = 16 * (m >> 2) + 4 * g + (m & 3);
    float dv[3], dgam[3], dbet[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pS[c] = acm_group_sum<64>(pS[c]);
        const float v = hl[c * 64 + mycol], gm = hl[192 + c * 64 + mycol], bt = hl[384 + c * 64 + mycol];
        dv[c] = fmaf(gm, pA[c], bt * pS[c]);
        dgam[c] = v * pA[c];
        dbet[c] = v * pS[c];
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) dmix[q] = acm_group_sum<64>(dmix[q]);
    __syncthreads();                               // every wave is done with the staged parameters
    float* slab = lds + wv * NPG;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        slab[(0 * 3 + c) * 64 + mycol] = dv[c];
        slab[(1 * 3 + c) * 64 + mycol] = dgam[c];
        slab[(2 * 3 + c) * 64 + mycol] = dbet[c];
    }
    if (lane < 9) {
        float v = dmix[0];
#pragma unroll
        for (int q = 1; q < 9; ++q) v = lane == q ? dmix[q] : v;
        slab[9 * 64 + lane] = v;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < NPG; q += 256)
        partial[(long)blockIdx.x * NPG + q] = (lds[q] + lds[NPG + q]) + (lds[2 * NPG + q] + lds[3 * NPG + q]);
}

}  // namespace

// Returns the number of blocks launched (> 0), 0 when the configuration is not this kernel's (the caller runs the older
// kernels), or a negative acm_status_t.  partial: max_blocks x (3 * 3 * 64 + 9) floats.
int acm_bwd_local16(const acm_conv_bwd_local_t* p, int64_t n_rows, float* partial, int max_blocks, hipStream_t s) {
    if (p->f_out != 64 || p->n_channels != 3 || p->post_scale || n_rows < 1 || !(acm_tuning().rows16 & ACM_ROWS16_LOCAL)) return 0;
    for (const void* q : {(const void*)p->pre, (const void*)p->s_mlp, (const void*)p->grad_out, (const void*)p->g_low,
                          (const void*)p->g_high, (const void*)p->g_mlp})
        if (((uintptr_t)q) % 16 != 0) return 0;
    int64_t ld_max = 0;
    for (int64_t ld : {p->ld_pre, p->ld_s_mlp, p->ld_grad_out, p->ld_g_low, p->ld_g_high, p->ld_g_mlp}) {
        if (ld % 4 != 0) return 0;
        ld_max = ld > ld_max ? ld : ld_max;
    }
    if (n_rows * ld_max >= (int64_t)INT32_MAX) return 0;                   // 32-bit element offsets
    int grid = (int)((n_rows + 63) / 64);
    if (grid > 512) grid = 512;                    // two resident workgroups per CU (220 registers); fewer slabs for the flush
    if (grid > max_blocks) grid = max_blocks;
    if (p->layernorm) hipLaunchKernelGGL((bwd_local16_kernel<true>), dim3(grid), dim3(256), 0, s, *p, (int)n_rows, partial);
    else hipLaunchKernelGGL((bwd_local16_kernel<false>), dim3(grid), dim3(256), 0, s, *p, (int)n_rows, partial);
    if (hipGetLastError() != hipSuccess) return -ACM_EHIP;
    return grid;
}
