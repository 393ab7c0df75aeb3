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
#include "acm_stream_device.h"
#include "acm_rows16_device.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ACM_E16_NEXT_LDS (64 * 8)


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

// ---------------------------------------------------------------- backward (K3a) in the same layout
// Per wave step (16 rows): the projections again on the matrix pipe (H is not stored: 768 B per row), the row's head
// statistics as the forward computed them (head_stats), the head backward with one cross-row sum per reduction, and
//     dW_c[f][col] += sum_rows A_c[row][f] G_c[row][col]
// on the matrix pipe as before -- but its operands want the ROW index on lane >> 4 (the contraction index of
// v_mfma_f32_16x16x4_f32) while the transposed layout has it on lane & 15, so each channel's G passes through a per-wave
// LDS tile (16 rows x 64 columns, written as 16-byte pieces, read back as the MFMA's B operand; the wave's own LDS
// accesses are ordered, no barrier).  The row-sums of the head parameters (A_c[col] = sum_rows ds_c xhat_c, see
// row_channel_backward) accumulate per lane for the lane's own row and are summed over the 16 row-lanes once, after the
// row loop.
#define ACM_B16_TS 68                  /* floats per tile row: 16-byte writes of eight consecutive row-lanes and the B-operand reads
                                          (rows 4 g + s: two row groups per LDS pass, 16 banks apart) are conflict-free */
#define ACM_B16_PS 17                  /* floats per [P | x] row */
#define ACM_B16_LDS (4 * (2144 + 384) + 64)    /* tiles | [P|x] rows | head parameters | weights; the end-of-kernel slabs alias it */

// PROJ: the following layer's projection backward rides along (acm_conv_agg_bwd_t.proj_*): grad_out is formed per row from
// proj_dz (6 floats) and the 64 x 6 weight table in LDS instead of being read (256 B per row), and proj_d_w = out^T proj_dz
// is one more set of MFMAs over the `out` tile (which passes through the same LDS tile as the G_c do).
// GATHER: a workgroup of EIGHT waves, one per SIMD for each role: waves 0-3 run this backward, waves 4-7 walk the operator's
// id streams for the next training step's P = A_low dropout(x) (acm_conv_agg_bwd_t.next_agg; stream_gather_role).  The
// kernel is compiled for two waves per SIMD either way, so the pair costs the backward no occupancy; its waves get the
// vector and matrix pipes almost to themselves (the gather waves wait on memory), the gather waves the memory system.
template <bool LN, bool OUT_MASK, bool PROJ, bool GATHER>
__device__ __forceinline__ void bwd16_body(const acm_conv_agg_bwd_t& p, int n_rows, float* __restrict__ partial, const GatherRole* gr) {
    __shared__ __attribute__((aligned(16))) float lds[ACM_B16_LDS];
    const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    float* gt = lds + wv * (16 * ACM_B16_TS);                 // this wave's G tile
    float* px = lds + 4 * 16 * ACM_B16_TS + wv * 16 * ACM_B16_PS;   // this wave's [P | x] rows, 16 floats each
    float* hl = lds + 4 * 16 * ACM_B16_TS + 4 * 16 * ACM_B16_PS + 16;   // [att_vec | gamma | beta][c][col]  (576 floats) + u_c (192); 16-byte aligned
    float* ul = hl + 576;
    float* wl = ul + 192;                                     // A operands of the projections, [(c, kb, t)][lane]
    float* pw = wl + 24 * 64;                                 // PROJ: [col][8] = [W_L'(col, :) | W_H'(col, :) | W_I'(col, :) | 0]
    static_assert(ACM_B16_LDS >= 4 * 16 * ACM_B16_TS + 4 * 16 * ACM_B16_PS + 16 + 768 + 24 * 64 + 512, "tiles | [P|x] rows | head parameters | weights");
    static_assert(ACM_B16_LDS >= 4 * (2144 + 384), "the four end-of-kernel slabs (f_in = 8: 2121 -> 2144 entries, proj_f = 2) alias everything");
    const int f_in = p.f_in;
    // one round of independent global loads, one barrier: the head parameters, u = att_vec * gamma, the projections' A
    // operands W_c[f = 4 kb + g][col = 16 t + m] (the same for every wave), c1_c = mean_col(u_c)
    for (int idx = threadIdx.x; idx < 576; idx += blockDim.x) {
        const int arr = idx / 192, c = (idx / 64) % 3, col = idx & 63;
        const float* av = c == 0 ? p.att_vec[0] : (c == 1 ? p.att_vec[1] : p.att_vec[2]);
        float v;
        if (arr == 0) v = av[col];
        else if (LN) {
            const float* gw = c == 0 ? p.ln_weight[0] : (c == 1 ? p.ln_weight[1] : p.ln_weight[2]);
            const float* gb = c == 0 ? p.ln_bias[0] : (c == 1 ? p.ln_bias[1] : p.ln_bias[2]);
            v = arr == 1 ? gw[col] : gb[col];
        } else v = arr == 1 ? 1.f : 0.f;
        hl[idx] = v;
    }
    float c1[3];
    if (threadIdx.x < 192) {
        const int c = threadIdx.x >> 6, col = threadIdx.x & 63;
        const float* av = c == 0 ? p.att_vec[0] : (c == 1 ? p.att_vec[1] : p.att_vec[2]);
        float u = av[col];
        if (LN) {
            const float* gw = c == 0 ? p.ln_weight[0] : (c == 1 ? p.ln_weight[1] : p.ln_weight[2]);
            u *= gw[col];
        }
        ul[threadIdx.x] = u;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float u = p.att_vec[c][lane];
        if (LN) u *= p.ln_weight[c][lane];
        c1[c] = acm_group_sum<64>(u) * (1.0f / 64.0f);
    }
    for (int idx = threadIdx.x; idx < 24 * 64; idx += blockDim.x) {
        const int e = idx >> 6, l2 = idx & 63, c = e >> 3, kb = (e >> 2) & 1, t = e & 3, f = 4 * kb + (l2 >> 4);
        const float* w = c == 0 ? p.w_low : (c == 1 ? p.w_high : p.w_mlp);
        wl[idx] = f < f_in ? w[(long)f * p.ld_w + 16 * t + (l2 & 15)] : 0.f;
    }
    const int nq = PROJ ? 3 * p.proj_f : 0;                   // columns of proj_dz
    if (PROJ) {
        for (int idx = threadIdx.x; idx < 512; idx += blockDim.x) {
            const int col = idx >> 3, j = idx & 7, c = j / p.proj_f, q = j % p.proj_f;
            const float* w = c == 0 ? p.proj_w_low : (c == 1 ? p.proj_w_high : p.proj_w_mlp);
            pw[idx] = (c < 3) ? w[(long)col * p.proj_ld_w + q] : 0.f;
        }
    }
    float mixm[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) mixm[q] = p.att_mix[q];
    __syncthreads();
    if (GATHER && wv >= 4) {                      // the gather role; then the same two barriers as the backward's end phase
        stream_gather_role(*gr, __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (wv - 4)));
        __syncthreads();
        __syncthreads();
        return;
    }
    const float lo_a = p.relu_after ? 0.f : -INFINITY, lo_m = p.relu_mlp ? 0.f : -INFINITY;
    constexpr bool out_mask = OUT_MASK;
    const float post_gain = p.post_drop.p > 0.f ? 1.0f / (1.0f - p.post_drop.p) : 1.f;
    const unsigned ld_agg = (unsigned)p.ld_agg, ld_xs = (unsigned)p.ld_xs, ld_go = (unsigned)p.ld_grad_out,
                   ld_out = (unsigned)p.ld_out, ld_hs = (unsigned)p.ld_head_stats;
    const int wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    const float wq = g == 0 ? 1.f : 0.f;            // a row's scalars sit in four lanes: one of them accumulates
    const float fsel = m < 8 ? 1.f : 0.f;           // A operand of the dW products: feature f = m (< f_pad)

    f32x4 acc[3][4], acc2[4];            // acc2 (PROJ): proj_d_w^T tiles, D[i = j of proj_dz][col]
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned ld_pz = (unsigned)p.ld_proj_dz;
    float pA[3], pS[3], dmix[9];         // pA[c]: lane (g, m) accumulates column 16 (m >> 2) + 4 g + (m & 3) of A_c
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pS[c] = pA[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) dmix[q] = 0.f;

    int base = wave * 16;
    float nPa = 0.f, nPb = 0.f, nxa = 0.f, nxb = 0.f;
    f32x4 ngo[4], nou[4], nst[3];
    // the operands of the projections (P, x: 64 B per row) are requested one step ahead; grad_out / out (512 B per row), the
    // head statistics and proj_dz at the top of their own step -- the projections' MFMAs run while they arrive
#define ACM_B16_LOAD(BASE)                                                                              \
    do {                                                                                                \
        const unsigned r2 = (unsigned)min((BASE) + m, n_rows - 1);                                      \
        nPa = p.agg[r2 * ld_agg + g], nPb = p.agg[r2 * ld_agg + 4 + g];                                 \
        nxa = p.xs[r2 * ld_xs + g], nxb = p.xs[r2 * ld_xs + 4 + g];                                     \
    } while (0)
    if (base < n_rows) ACM_B16_LOAD(base);
    for (; base < n_rows; base += nwaves * 16) {
        const bool valid = base + m < n_rows;
        const float Pa = nPa, Pb = nPb, xa = nxa, xb = nxb;
        float dzr[6], dza[4];            // PROJ: proj_dz of row m; of rows 4 g + s at column m (the A operand of proj_d_w)
        {
            const unsigned r1 = (unsigned)min(base + m, n_rows - 1);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!PROJ) ngo[t] = *reinterpret_cast<const f32x4*>(p.grad_out + r1 * ld_go + 16 * t + 4 * g);
                if (out_mask || PROJ) nou[t] = *reinterpret_cast<const f32x4*>(p.out + r1 * ld_out + 16 * t + 4 * g);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) nst[q] = *reinterpret_cast<const f32x4*>(p.head_stats + r1 * ld_hs + 4 * q);
            if (PROJ) {
#pragma unroll
                for (int j = 0; j < 6; ++j) dzr[j] = (j < nq && valid) ? p.proj_dz[r1 * ld_pz + j] : 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int rs = base + 4 * g + s;
                    dza[s] = (m < nq && rs < n_rows) ? p.proj_dz[(unsigned)min(rs, n_rows - 1) * ld_pz + min(m, 5)] : 0.f;
                }
            }
        }
        ACM_B16_LOAD(base + nwaves * 16);           // the next step's operands (the addresses are clamped)
        const int gq = acm_opaque(g), mq = acm_opaque(m);
        px[mq * ACM_B16_PS + gq] = Pa, px[mq * ACM_B16_PS + 4 + gq] = Pb, px[mq * ACM_B16_PS + 8 + gq] = xa, px[mq * ACM_B16_PS + 12 + gq] = xb;
        const float opa[3] = {Pa, xa - Pa, xa}, opb[3] = {Pb, xb - Pb, xb};
        f32x4 D[3][4];
        const int lq = acm_opaque(lane);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                D[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[((c * 2 + 0) * 4 + t) * 64 + lq], opa[c], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                D[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[((c * 2 + 1) * 4 + t) * 64 + lq], opb[c], D[c][t], 0, 0, 0);
        f32x4 dO[4];
        const float gate = valid ? post_gain : 0.f;
        if (PROJ) {
            // the following layer's weight gradient: out^T proj_dz, the `out` tile through LDS like the G_c below; then this
            // layer's output gradient row by row, proj_dz[row] [W_L' | W_H' | W_I']^T, masked by the post-op right away
#pragma unroll
            for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(gt + mq * ACM_B16_TS + 16 * t + 4 * gq) = nou[t];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(dza[s], gt[(4 * gq + s) * ACM_B16_TS + 16 * t + mq], acc2[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 wa = *reinterpret_cast<const float4*>(pw + (16 * t + 4 * gq + r) * 8);
                    const float2 wb = *reinterpret_cast<const float2*>(pw + (16 * t + 4 * gq + r) * 8 + 4);
                    float v = dzr[0] * wa.x;
                    v = fmaf(dzr[1], wa.y, v), v = fmaf(dzr[2], wa.z, v), v = fmaf(dzr[3], wa.w, v);
                    v = fmaf(dzr[4], wb.x, v), v = fmaf(dzr[5], wb.y, v);
                    dO[t][r] = out_mask ? ((nou[t][r] != 0.f) ? v * gate : 0.f) : v;        // (dzr is zero on padding rows)
                }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dO[t][r] = out_mask ? ((nou[t][r] != 0.f) ? ngo[t][r] * gate : 0.f) : (valid ? ngo[t][r] : 0.f);
        }
        const float mean[3] = {nst[0][0], nst[0][1], nst[0][2]}, rstd[3] = {nst[0][3], nst[1][0], nst[1][1]};
        const float gsig[3] = {nst[1][2], nst[1][3], nst[2][0]}, al[3] = {nst[2][1], nst[2][2], nst[2][3]};
        // ---- mix / softmax / sigmoid backward: ds_c = dL/ds_c per row
        float dal[3], ds[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float lo = c < 2 ? lo_a : lo_m;
            float part = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    D[c][t][r] = fmaxf(D[c][t][r], lo);
                    part = fmaf(dO[t][r], D[c][t][r], part);
                }
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
        // ---- one channel at a time: G_c -> LDS tile -> dW_c on the matrix pipe
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float lo = c < 2 ? lo_a : lo_m;
            const float aal = p.scale * al[c];
            // two passes over the lane's 16 columns, four at a time, so that neither xhat nor G is held as a whole: (A) the
            // row sums (contrib -> the reduce-scatter; t2 = sum_col u xhat), (B) xhat again, G, straight into the LDS tile
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
                    G[r] = D[c][t][r] > lo ? v : 0.f;
                }
                *reinterpret_cast<f32x4*>(gt + mq * ACM_B16_TS + 16 * t + 4 * gq) = G;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // A operand: feature m of row 4 g + s of this wave step (P | x rows parked in LDS at the top of the step)
                const float ap = px[(4 * gq + s) * ACM_B16_PS + (mq & 7)], ax = px[(4 * gq + s) * ACM_B16_PS + 8 + (mq & 7)];
                const float aop = fsel * (c == 0 ? ap : (c == 1 ? ax - ap : ax));
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, gt[(4 * gq + s) * ACM_B16_TS + 16 * t + mq], acc[c][t], 0, 0, 0);
            }
        }
    }
#undef ACM_B16_LOAD
    // ---- end of the row loop: head-parameter sums over the 16 row-lanes, then the block's partial slab
    const int npg0 = 3 * f_in * 64 + 9 * 64 + 9;
    const int off2 = (npg0 + 31) & ~31;              // PROJ: proj_d_w behind d_params at a whole group of 32 (the second phase
    const int npg = PROJ ? off2 + 64 * nq : npg0;    // sums it by lines), [c][col][q] = flat index (j / f') 64 f' + col f' + j % f'
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
    __syncthreads();                               // every wave is done with the tiles and the staged parameters
    float* slab = lds + wv * npg;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 4 * g + r;
                if (f < f_in) slab[(c * f_in + f) * 64 + 16 * t + m] = acc[c][t][r];
            }
    {
        const int b2 = 3 * f_in * 64;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            slab[b2 + (0 * 3 + c) * 64 + mycol] = dv[c];
            slab[b2 + (1 * 3 + c) * 64 + mycol] = dgam[c];
            slab[b2 + (2 * 3 + c) * 64 + mycol] = dbet[c];
        }
    }
    if (lane < 9) {
        float v = dmix[0];
#pragma unroll
        for (int q = 1; q < 9; ++q) v = lane == q ? dmix[q] : v;
        slab[3 * f_in * 64 + 9 * 64 + lane] = v;
    }
    if (PROJ) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 4 * g + r, col = 16 * t + m;
                if (j < nq) slab[off2 + (j / p.proj_f) * 64 * p.proj_f + col * p.proj_f + j % p.proj_f] = acc2[t][r];
            }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < npg; q += 256) {                    // (the four backward waves)
        float v = (lds[q] + lds[npg + q]) + (lds[2 * npg + q] + lds[3 * npg + q]);
        if (PROJ && q >= npg0 && q < off2) v = 0.f;                    // the gap up to the group boundary
        partial[((long)(q >> 5) * gridDim.x + blockIdx.x) * 32 + (q & 31)] = v;
    }
}

template <bool LN, bool OUT_MASK, bool PROJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void agg_bwd16_kernel(acm_conv_agg_bwd_t p, int n_rows, float* __restrict__ partial) {
    bwd16_body<LN, OUT_MASK, PROJ, false>(p, n_rows, partial, nullptr);
}
template <bool LN, bool OUT_MASK, bool PROJ>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void agg_bwd16_gather_kernel(acm_conv_agg_bwd_t p, int n_rows, float* __restrict__ partial,
                                                                                                   GatherRole gr) {
    bwd16_body<LN, OUT_MASK, PROJ, true>(p, n_rows, partial, &gr);
}

}  // namespace

// The row-local forward stage over an existing P = A_low X (p->agg).  Returns ACM_OK after a launch, -1 when the
// configuration is not the one this kernel is written for (the caller then runs agg_epilogue_kernel), or an error.
int acm_agg_epi16(const acm_conv_agg_fwd_t* p, int64_t n_rows, bool* next_done, hipStream_t s) {
    *next_done = false;
    if (p->n_channels != 3 || p->f_pad != 8 || p->f_out != 64 || !(acm_tuning().rows16 & ACM_ROWS16_EPI)) return -1;
    int64_t ld_max = 64;
    for (int64_t ld : {p->ld_agg, p->ld_xs, p->ld_out, p->ld_head_stats, p->ld_post_scale, p->ld_agg_copy, p->ld_xs_copy})
        ld_max = ld > ld_max ? ld : ld_max;
    if (n_rows * ld_max >= (int64_t)INT32_MAX) return -1;                  // 32-bit element offsets
    if ((((uintptr_t)p->out) % 16) != 0 || (p->ld_out % 4) != 0) return -1;
    if (p->post_scale && ((((uintptr_t)p->post_scale) % 16) != 0 || (p->ld_post_scale % 4) != 0)) return -1;
    const bool next = p->next_f > 0;
    int grid = (int)((n_rows + 63) / 64);
    if (grid > 1024) grid = 1024;                    // four workgroups (sixteen waves) per CU
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
    ACM_E16(agg_epi16_kernel);
#undef ACM_E16
    ACM_CHECK_HIP(hipGetLastError());
    *next_done = next;
    return ACM_OK;
}

// The row-local backward over the forward's head_stats.  Returns the number of blocks launched (> 0), 0 when the
// configuration is not this kernel's (the caller runs agg_bwd_kernel), or a negative acm_status_t.  The partial slab has
// npg + 64 * 3 * proj_f entries per block (acm_conv_agg_bwd_t.proj_*: the following layer's weight gradient behind d_params).
int acm_agg_bwd16(const acm_conv_agg_bwd_t* p, int64_t n_rows, float* partial, int max_blocks, hipStream_t s, const GatherRole* gr,
                  int gather_blocks) {
    if (p->n_channels != 3 || p->f_pad != 8 || p->f_out != 64 || !p->head_stats || !(acm_tuning().rows16 & ACM_ROWS16_BWD)) return 0;
    const bool out_mask = p->out != nullptr && p->post_relu && !p->post_scale;
    const bool no_post = !p->post_relu && !p->post_scale && !(p->post_drop.p > 0.f);
    if (!out_mask && !no_post) return 0;
    const bool proj = p->proj_dz != nullptr;
    if (proj && (!p->out || p->proj_f < 1 || p->proj_f > 2 || !p->proj_w_low || !p->proj_w_high || !p->proj_w_mlp || !p->proj_d_w ||
                 p->ld_proj_dz < 3 * p->proj_f || p->proj_ld_w < p->proj_f || n_rows * p->ld_proj_dz >= (int64_t)INT32_MAX))
        return 0;
    for (const void* q : {(const void*)(proj ? nullptr : p->grad_out), (const void*)p->out, (const void*)p->head_stats})
        if (((uintptr_t)q) % 16 != 0) return 0;
    if ((!proj && p->ld_grad_out % 4 != 0) || ((out_mask || proj) && p->ld_out % 4 != 0) || p->ld_head_stats % 4 != 0) return 0;
    acm_conv_agg_bwd_t q = *p;
    if (!out_mask && !proj) q.out = nullptr;
    int grid = (int)((n_rows + 63) / 64);
    int cap = 512;
    if (cap > max_blocks) cap = max_blocks;
    if (grid > cap) grid = cap;
    if (gr) {                                         // one workgroup of eight waves per four stream waves
        if (gather_blocks < 1 || gather_blocks > max_blocks) return 0;
        grid = gather_blocks;
    }
#define ACM_B16(LNv, OMv, PJv)                                                                                                      \
    do {                                                                                                                            \
        if (gr) hipLaunchKernelGGL((agg_bwd16_gather_kernel<LNv, OMv, PJv>), dim3(grid), dim3(512), 0, s, q, (int)n_rows, partial, *gr); \
        else hipLaunchKernelGGL((agg_bwd16_kernel<LNv, OMv, PJv>), dim3(grid), dim3(256), 0, s, q, (int)n_rows, partial);            \
    } while (0)
    if (proj) {                                       // (the `out` tile is read either way: the mask flag only gates its use)
        if (p->layernorm) { if (out_mask) ACM_B16(true, true, true); else ACM_B16(true, false, true); }
        else { if (out_mask) ACM_B16(false, true, true); else ACM_B16(false, false, true); }
    } else {
        if (p->layernorm) { if (out_mask) ACM_B16(true, true, false); else ACM_B16(true, false, false); }
        else { if (out_mask) ACM_B16(false, true, false); else ACM_B16(false, false, false); }
    }
#undef ACM_B16
    if (hipGetLastError() != hipSuccess) return -ACM_EHIP;
    return grid;
}
