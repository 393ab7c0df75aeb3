// The aggregate-first ACM layer for a WIDE dense input (16 < F_in <= 128, hidden width 64; ACM-Geometric/layers.py:101-108 with
// arXiv-year's 128 or pokec's 65 input features), after P = A_low Xd has been gathered: projections, LayerNorm / attention
// head, mix and post-op in ONE row-local kernel (round 5).
//
// Before: two split-bf16 products ([P W_L | P W_H], [Xd W_H | Xd W_I]: 2 x 49 us on the arXiv-year-shaped graph) wrote 2 x 87 MB
// that the head kernel (84 us) read back.  Here a wave takes sixteen rows of P and Xd straight into operand registers, forms
// the three operands P, Xd - P, Xd (as the narrow aggregate-first kernel does, acm_conv_agg16.hip), splits each into three bf16
// vectors and runs the 6-MFMA step of acm_gemm_bx3.hip against W_L, W_H, W_I, which sit in LDS for the whole launch already
// split and laid out as A operands (3 parts x 12 column tiles x 4 k blocks x 1 KB = 144 KB).  The roles are transposed (A = W^T,
// B = rows^T), so lane (g, m) ends up with columns 16 t + 4 g + r of row m for the three channels -- the layout of the
// sixteen-rows head (acm_rows16_device.h): a head reduction is 15 in-lane adds and one cross-row sum for 16 rows at once.
// The pre-activations (pre_L, pre_H, Z_I) leave as 16-byte stores for the backward (acm_conv_bwd_local), the output row
// after mix, ReLU and dropout.  Stream: 2 x 4 K in, 4 (64 + 128 + 64 + 4) out per row.
#include "acm_conv_device.h"
#include "acm_rows16_device.h"
#include "acm_bx3_device.h"

namespace {

constexpr int AW_TILES = 12;              // 3 channels x 4 column tiles of 16

// first column of X behind k block kb (see bx3_k: kb 0, 1 -> columns 0..63, kb 2, 3 -> 64..127)
__device__ __forceinline__ int aw_kb_first(int kb) { return 64 * (kb >> 1) + 32 * (kb & 1); }

template <bool LN, int NKB>
__global__ __launch_bounds__(512, 2) void aggw_head_kernel(int n_rows, int K, int f_in, const float* __restrict__ agg, long ld_agg,
                                                           const float* __restrict__ xs, long ld_xs, const float* __restrict__ w_low,
                                                           const float* __restrict__ w_high, const float* __restrict__ w_mlp, long ld_w,
                                                           float* __restrict__ zi, long ld_zi, acm_conv_fwd_t p) {
    constexpr int n_kb = NKB;                      // k blocks of 32 input columns in use (compile time: every LDS operand read is
                                                   // one base register + an immediate offset)
    extern __shared__ __attribute__((aligned(16))) u32x4 Ws[];       // [part 3][tile 12][kb n_kb][lane 64] | u[3][64] floats
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    float* ulds = reinterpret_cast<float*>(Ws + 3 * AW_TILES * n_kb * 64);
    // W_c^T as A operands: lane (gi, i) of tile j = 4 c + t, k block kb holds W_c[bx3_k(kb, gi, e)][16 t + i], e = 0..7.
    // Every load of the block is in flight before the first split (unconditional loads from a clamped row: a guarded load
    // makes the compiler wait for each one -- 48 dependent round trips before the first row panel).
    {
        constexpr int WIT = (AW_TILES * NKB * 64 + 511) / 512;
        float wst[WIT][8];
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int idx = min((int)threadIdx.x + 512 * it, AW_TILES * NKB * 64 - 1);
            const int ln = idx & 63, kb = (idx >> 6) % NKB, j = (idx >> 6) / NKB, gi = ln >> 4, col = 16 * (j & 3) + (ln & 15);
            const float* w = (j >> 2) == 0 ? w_low : ((j >> 2) == 1 ? w_high : w_mlp);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kr = bx3_k(kb, gi, e);
                const float v = w[(long)min(kr, f_in - 1) * ld_w + col];
                wst[it][e] = kr < f_in ? v : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int idx = threadIdx.x + 512 * it;
            const int ln = idx & 63, kb = (idx >> 6) % NKB, j = (idx >> 6) / NKB;
            u32x4 h, mdl, l;
            split3(wst[it], h, mdl, l);
            if (idx < AW_TILES * NKB * 64) {
                Ws[((0 * AW_TILES + j) * NKB + kb) * 64 + ln] = h;
                Ws[((1 * AW_TILES + j) * NKB + kb) * 64 + ln] = mdl;
                Ws[((2 * AW_TILES + j) * NKB + kb) * 64 + ln] = l;
            }
        }
    }
    // u_c = gamma_c (.) att_vec_c (LayerNorm folded into the attention vector, as in acm_conv_agg16.hip)
#pragma unroll
    for (int c = 0; c < 3; ++c) {                  // (compile-time channel indices only: see acm_conv_agg16.hip)
        if (wv == c) {
            float u = p.att_vec[c][lane];
            if (LN) u *= p.ln_weight[c][lane];
            ulds[c * 64 + lane] = u;
        }
    }
    float c0[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) c0[c] = LN ? acm_group_sum<64>(p.ln_bias[c][lane] * p.att_vec[c][lane]) : 0.f;
    float mixm[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) mixm[q] = p.att_mix[q];
    const AcmDropCtx dc = acm_drop_ctx(p.post_drop);
    const float lo_a = p.relu_after ? 0.f : -INFINITY, lo_m = p.relu_mlp ? 0.f : -INFINITY;
    const float lo_post = p.post_relu ? 0.f : -INFINITY;
    const int npan = (n_rows + 15) / 16, stride = gridDim.x * 8;

    f32x4 nP[2][4], nX[2][4];
    // half G of a panel's rows: columns 64 G .. 64 G + 63 (k blocks 2 G, 2 G + 1)
    auto fetch_half = [&](int pan, int G) {
        int row = pan * 16 + m;
        row = row < n_rows ? row : n_rows - 1;                        // (results of such rows are not stored)
        const float* ap = agg + (long)row * ld_agg;
        const float* xp = xs + (long)row * ld_xs;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // (pad columns beyond K are never multiplied by a nonzero weight row: they are read from the last valid
            //  16-byte block instead of being branched around)
            const int off = min(64 * G + 16 * q + 4 * g, K - 4);
            nP[G][q] = *reinterpret_cast<const f32x4*>(ap + off);
            nX[G][q] = *reinterpret_cast<const f32x4*>(xp + off);
        }
    };
    int pan = blockIdx.x * 8 + wv;
    if (pan < npan) {
        fetch_half(pan, 0);
        if (NKB > 2) fetch_half(pan, 1);
    }
    __syncthreads();                               // W and u are in LDS
    for (; pan < npan; pan += stride) {
        f32x4 D[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) D[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kb >= NKB) continue;
            // operands of the three channels: P, Xd - P, Xd
            u32x4 oh[3], om[3], ol[3];
            {
                const f32x4 pu = nP[kb >> 1][2 * (kb & 1)], pv = nP[kb >> 1][2 * (kb & 1) + 1];
                const f32x4 xu = nX[kb >> 1][2 * (kb & 1)], xv = nX[kb >> 1][2 * (kb & 1) + 1];
                const float a8[8] = {pu[0], pu[1], pu[2], pu[3], pv[0], pv[1], pv[2], pv[3]};
                const float b8[8] = {xu[0] - pu[0], xu[1] - pu[1], xu[2] - pu[2], xu[3] - pu[3],
                                     xv[0] - pv[0], xv[1] - pv[1], xv[2] - pv[2], xv[3] - pv[3]};
                const float c8[8] = {xu[0], xu[1], xu[2], xu[3], xv[0], xv[1], xv[2], xv[3]};
                split3(a8, oh[0], om[0], ol[0]);
                split3(b8, oh[1], om[1], ol[1]);
                split3(c8, oh[2], om[2], ol[2]);
            }
            // a step = two column tiles of one channel, their accumulators alternating (small terms first); the W operands of
            // the next step are read while this one feeds the matrix pipe (as gemm_bx3_nn_kernel)
            u32x4 wbuf[2][2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int part = 0; part < 3; ++part) wbuf[0][t][part] = Ws[((part * AW_TILES + t) * n_kb + kb) * 64 + lane];
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                if (s + 1 < 6) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int part = 0; part < 3; ++part)
                            wbuf[(s + 1) & 1][t][part] = Ws[((part * AW_TILES + 2 * (s + 1) + t) * n_kb + kb) * 64 + lane];
                }
                const int c = s >> 1, t0 = 2 * (s & 1), t1 = t0 + 1;
                const u32x4 ah = wbuf[s & 1][0][0], am = wbuf[s & 1][0][1], al = wbuf[s & 1][0][2];
                const u32x4 bh = wbuf[s & 1][1][0], bm = wbuf[s & 1][1][1], bl = wbuf[s & 1][1][2];
                D[c][t0] = mma(al, oh[c], D[c][t0]);
                D[c][t1] = mma(bl, oh[c], D[c][t1]);
                D[c][t0] = mma(ah, ol[c], D[c][t0]);
                D[c][t1] = mma(bh, ol[c], D[c][t1]);
                D[c][t0] = mma(am, om[c], D[c][t0]);
                D[c][t1] = mma(bm, om[c], D[c][t1]);
                D[c][t0] = mma(am, oh[c], D[c][t0]);
                D[c][t1] = mma(bm, oh[c], D[c][t1]);
                D[c][t0] = mma(ah, om[c], D[c][t0]);
                D[c][t1] = mma(bh, om[c], D[c][t1]);
                D[c][t0] = mma(ah, oh[c], D[c][t0]);
                D[c][t1] = mma(bh, oh[c], D[c][t1]);
            }
            // the half of the operand rows that has just been consumed is requested for the NEXT panel at once: those loads
            // travel during the rest of this panel's products and its head (one register set, not two)
            if (kb == 1 || kb == NKB - 1) fetch_half(pan + stride < npan ? pan + stride : pan, kb >> 1);
        }
        const int row = pan * 16 + m;
        const bool valid = row < n_rows;
        const long rr = valid ? row : n_rows - 1;
        // the pre-activations the backward reads: [pre_L | pre_H] and Z_I, before any ReLU
        if (valid) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                *reinterpret_cast<f32x4*>(p.pre + rr * p.ld_pre + 16 * t + 4 * g) = D[0][t];
                *reinterpret_cast<f32x4*>(p.pre + rr * p.ld_pre + 64 + 16 * t + 4 * g) = D[1][t];
                *reinterpret_cast<f32x4*>(zi + rr * ld_zi + 16 * t + 4 * g) = D[2][t];
            }
        }
        // ---- head: statistics and attention scalars of row m (four lanes per row), as acm_conv_agg16.hip: epi16_body
        const int gq = acm_opaque(g);
        float rstd[3], gs[3];
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
                rstd[c] = acm_rsqrt(row4_sum(q) * (1.0f / 64.0f) + ACM_LN_EPS);
                dot = fmaf(rstd[c], row4_sum(dot), c0[c]);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(ulds + c * 64 + 16 * t + 4 * gq);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dot = fmaf(D[c][t][r], u[r], dot);
                }
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
        if (valid && g == 1) *reinterpret_cast<float4*>(p.att + (size_t)rr * 4) = make_float4(al[0], al[1], al[2], 0.f);
        // ---- mix, post-op, store
        const float a0 = al[0] * p.scale, a1 = al[1] * p.scale, a2 = al[2] * p.scale;
        f32x4 o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[t][r] = fmaxf(fmaf(a2, D[2][t][r], fmaf(a1, D[1][t][r], a0 * D[0][t][r])), lo_post);
        if (p.post_scale) {
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] *= *reinterpret_cast<const f32x4*>(p.post_scale + rr * p.ld_post_scale + 16 * t + 4 * g);
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
            for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(p.out + rr * p.ld_out + 16 * t + 4 * g) = o[t];
        }
    }
}

}  // namespace

extern "C" int acm_conv_aggw_fwd(int64_t n_rows, int64_t f_in, int64_t f_pad, const float* agg, int64_t ld_agg, const float* xs,
                                 int64_t ld_xs, const float* w_low, const float* w_high, const float* w_mlp, int64_t ld_w,
                                 float* zi, int64_t ld_zi, const acm_conv_fwd_t* p, acm_stream_t stream) {
    ACM_REQUIRE(p && agg && xs && w_low && w_high && w_mlp && zi, ACM_EINVAL, "acm_conv_aggw_fwd: NULL argument");
    ACM_REQUIRE(n_rows >= 0 && n_rows < INT32_MAX / 16, ACM_ESHAPE, "acm_conv_aggw_fwd: bad row count");
    ACM_REQUIRE(p->f_out == 64 && p->n_channels == 3 && f_in > 0 && f_in <= f_pad && f_pad <= 128 && f_pad % 4 == 0 && !p->gather_bf16,
                ACM_EUNSUPPORTED, "acm_conv_aggw_fwd: three fp32 channels of 64 columns, f_in <= f_pad <= 128, f_pad %% 4 == 0 (got F %d, k %d, "
                "f_in %lld, f_pad %lld)", p->f_out, p->n_channels, (long long)f_in, (long long)f_pad);
    ACM_REQUIRE(p->out && p->pre && p->att && p->att_mix && p->att_vec[0] && p->att_vec[1] && p->att_vec[2], ACM_EINVAL,
                "acm_conv_aggw_fwd: NULL pointer in the parameter block");
    ACM_REQUIRE(!p->layernorm || (p->ln_weight[0] && p->ln_weight[1] && p->ln_weight[2] && p->ln_bias[0] && p->ln_bias[1] && p->ln_bias[2]),
                ACM_EINVAL, "acm_conv_aggw_fwd: layernorm needs ln_weight / ln_bias");
    auto al16 = [](const void* q, int64_t ld) { return ((uintptr_t)q) % 16 == 0 && ld % 4 == 0; };
    ACM_REQUIRE(al16(agg, ld_agg) && al16(xs, ld_xs) && al16(zi, ld_zi) && al16(p->out, p->ld_out) && al16(p->pre, p->ld_pre)
                    && (!p->post_scale || al16(p->post_scale, p->ld_post_scale)) && ((uintptr_t)p->att) % 16 == 0,
                ACM_EUNSUPPORTED, "acm_conv_aggw_fwd: every row (agg, xs, zi, out, pre, post_scale, att) must be 16-byte aligned");
    if (n_rows == 0) return ACM_OK;
    hipStream_t st = (hipStream_t)stream;
    int n_kb = 0;
    for (int kb = 0; kb < 4; ++kb)
        if (64 * (kb >> 1) + 32 * (kb & 1) < f_pad) n_kb = kb + 1;
    const size_t lds = (size_t)3 * AW_TILES * n_kb * 64 * 16 + 3 * 64 * sizeof(float);
    const int64_t npan = (n_rows + 15) / 16;
    int grid = (int)((npan + 7) / 8);
    if (grid > 256) grid = 256;
#define ACM_AGGW(LNv, KBv)                                                                                                        \
    do {                                                                                                                          \
        ACM_CHECK_HIP(hipFuncSetAttribute((const void*)aggw_head_kernel<LNv, KBv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((aggw_head_kernel<LNv, KBv>), dim3(grid), dim3(512), lds, st, (int)n_rows, (int)f_pad, (int)f_in, agg,      \
                           (long)ld_agg, xs, (long)ld_xs, w_low, w_high, w_mlp, (long)ld_w, zi, (long)ld_zi, *p);                   \
    } while (0)
#define ACM_AGGW_KB(LNv)                    \
    switch (n_kb) {                         \
        case 1: ACM_AGGW(LNv, 1); break;    \
        case 2: ACM_AGGW(LNv, 2); break;    \
        case 3: ACM_AGGW(LNv, 3); break;    \
        default: ACM_AGGW(LNv, 4); break;   \
    }
    if (p->layernorm) {
        ACM_AGGW_KB(true)
    } else {
        ACM_AGGW_KB(false)
    }
#undef ACM_AGGW_KB
#undef ACM_AGGW
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}
