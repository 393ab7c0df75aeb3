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


// ================================================================================================ backward
// K3 (the head's backward: G_L, G_H, G_I per row) AND the three weight gradients
//     dW_L = P^T G_L,   dW_H = (Xd - P)^T G_H,   dW_I = Xd^T G_I
// in one kernel.  Before: acm_conv_bwd_local wrote [G_L | G_H | G_I] (768 B per row), two transposed split-bf16 products read it
// back, split it again and parked both operands in LDS (63 + 2 x 49 us + two reduction launches on the arXiv-year-shaped graph).
// Here a workgroup of eight waves walks slabs of 128 rows in two phases:
//   (1) every wave runs K3 on sixteen rows in the sixteen-rows layout (acm_conv_local16.hip), exchanges its G values with the
//       neighbour row's lane so that each lane holds TWO consecutive rows of two columns, splits them into three bf16 parts
//       and writes them to LDS as the B operands of v_mfma_f32_16x16x32_bf16 (lane (g, j): rows 8 g .. 8 g + 7 of column j);
//   (2) wave w owns input features 16 w .. 16 w + 15: it loads P and Xd of the slab's rows TRANSPOSED straight from global
//       memory (lane (g, i): feature 16 w + i of rows 8 g .. 8 g + 7 -- eight dword loads, 64 contiguous bytes per row and
//       wave), forms P, Xd - P, Xd, splits, and runs six MFMAs per (channel, column tile, 32 rows) into accumulators that live
//       for the whole launch: dW[c][16 w + 4 g + r][16 t + j].
// At the end a workgroup leaves ONE partial of the weight gradients (groups of 32 elements: whole 128-byte lines for the second
// phase) and one of the head-parameter sums; acm_reduce_emit finishes both (deferred: inside the optimizer's launch).
constexpr int AWB_NPG = 3 * 3 * 64 + 9;            // head-parameter partial vector (the layout of bwd_local16_kernel)
constexpr int AWB_SLAB = 64;                       // rows per slab = 4 producer waves x 16
constexpr int AWB_GS = 2 * 2 * 3 * AW_TILES * 64;  // u32x4 entries of the G operand buffers: [buffer 2][sub 2][part 3][tile 12][lane 64]

// four fp32 (two pairs) -> three x two dwords of packed bf16 (hi, mid, lo), x = hi + mid + lo exactly
__device__ __forceinline__ void split3_pairs(const float (&x)[4], unsigned (&hi)[2], unsigned (&mid)[2], unsigned (&lo)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float a = x[2 * t], b = x[2 * t + 1];
        const float ra = a - bitsf(fbits(a) & 0xFFFF0000u), rb = b - bitsf(fbits(b) & 0xFFFF0000u);
        const float sa = ra - bitsf(fbits(ra) & 0xFFFF0000u), sb = rb - bitsf(fbits(rb) & 0xFFFF0000u);
        hi[t] = pack_hi16(fbits(a), fbits(b));
        mid[t] = pack_hi16(fbits(ra), fbits(rb));
        lo[t] = pack_hi16(fbits(sa), fbits(sb));
    }
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the global loads in flight -- the NEXT slab's
// rows, requested just before it.
__device__ __forceinline__ void aw_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool LN, int KT>
__global__ __launch_bounds__(512, 2) void aggw_bwd_kernel(acm_conv_bwd_local_t p, int n_rows, int K, const float* __restrict__ agg,
                                                          long ld_agg, const float* __restrict__ xs, long ld_xs,
                                                          float* __restrict__ part_head, float* __restrict__ part_w) {
    extern __shared__ __attribute__((aligned(16))) u32x4 Gs[];       // G operands | hl[576] | ul[192]   (Gs aliased by the slabs at the end)
    float* hl = reinterpret_cast<float*>(Gs + AWB_GS);             // [att_vec | gamma | beta][c][col]
    float* ul = hl + 576;                                           // u_c = att_vec_c (.) gamma_c
    const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    for (int idx = threadIdx.x; idx < 576; idx += 512) {
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
    const float wq = g == 0 ? 1.f : 0.f;   // a row's scalars sit in four lanes: one of them accumulates
    // where this lane's G pairs go: rows (2 h, 2 h + 1) of the eight-row group g2 of sub-slab `sub`, columns 4 g + {0, 1} (even
    // row lanes) or 4 g + {2, 3} (odd row lanes)
    const int odd = m & 1, sub = (wv & 3) >> 1, g2 = 2 * (wv & 1) + (m >> 3), h = (m & 7) >> 1;
    unsigned* gsw0 = reinterpret_cast<unsigned*>(Gs) + ((sub * 3 * AW_TILES * 64) + 16 * g2 + 4 * g + 2 * odd) * 4 + h;
    const int nslab = (n_rows + AWB_SLAB - 1) / AWB_SLAB;
    const int my_slabs = blockIdx.x < nslab ? (nslab - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    // Waves 0-3 PRODUCE (K3 on sixteen rows each -> G operands of slab `it` in buffer it & 1), waves 4-7 CONSUME (the products of
    // slab it - 1 from the other buffer): every SIMD holds one wave of each kind, so the vector pipe (K3) and the matrix pipe run
    // side by side and each role's memory latency hides under the other's arithmetic.  One barrier per slab.
    float* lds = reinterpret_cast<float*>(Gs);
    // (two disjoint code paths, each with its own loop and ONE barrier per iteration: the producers' K3 state and the consumers'
    //  accumulators never live in the same wave -- declared in one loop they would cost 96 registers of scratch per lane)
    if (wv < 4) {
    float pA[3], pS[3], dmix[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) pS[c] = pA[c] = 0.f;
#pragma unroll
    for (int q = 0; q < 9; ++q) dmix[q] = 0.f;
        // (requesting a slab's rows one barrier ahead, as the consumers do, measured SLOWER here: 110 -> 130 us)
        f32x4 nD[3][4], ndO[4];
        auto request = [&](int slab_) {
            const long rq = min(slab_ * AWB_SLAB + 16 * wv + m, n_rows - 1);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                nD[0][t] = *reinterpret_cast<const f32x4*>(p.pre + rq * p.ld_pre + 16 * t + 4 * g);
                nD[1][t] = *reinterpret_cast<const f32x4*>(p.pre + rq * p.ld_pre + 64 + 16 * t + 4 * g);
                nD[2][t] = *reinterpret_cast<const f32x4*>(p.s_mlp + rq * p.ld_s_mlp + 16 * t + 4 * g);
                ndO[t] = *reinterpret_cast<const f32x4*>(p.grad_out + rq * p.ld_grad_out + 16 * t + 4 * g);
            }
        };
        for (int it = 0; it <= my_slabs; ++it) {
            if (it < my_slabs) {
        const int slab = blockIdx.x + it * gridDim.x;
        unsigned* gsw = gsw0 + (it & 1) * (2 * 3 * AW_TILES * 64 * 4);
        // ------------------------------------------------------------------ producer: K3 on rows slab * 64 + 16 wv + m
        const int row = slab * AWB_SLAB + 16 * wv + m;
        const bool valid = row < n_rows;
        const int gq = acm_opaque(g), mq = acm_opaque(m);
        request(slab);
        f32x4 D[3][4], dO[4];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) D[c][t] = nD[c][t];
#pragma unroll
        for (int t = 0; t < 4; ++t) dO[t] = ndO[t];
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
        // undo the forward's fused post-op on the incoming gradient: ReLU of the mixed row (recomputed), dropout (regenerated)
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
        // one channel at a time: row sums for the head-parameter gradients, then G_c -> pairs of rows -> bf16 parts -> LDS
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
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(ul + c * 64 + 16 * t + 4 * gq);
                float G[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v;
                    if (LN) {
                        const float xh = (D[c][t][r] - mean[c]) * rstd[c];
                        v = fmaf(aal, dO[t][r], rstd[c] * (fmaf(ds[c], u[r], -m1) - xh * m2));
                    } else {
                        v = fmaf(aal, dO[t][r], ds[c] * u[r]);
                    }
                    G[r] = (valid && D[c][t][r] > lo) ? v : 0.f;
                }
                // rows (m, m ^ 1) swap halves: the even row's lane ends up with both rows of columns r = 0, 1, the odd row's
                // lane with both rows of r = 2, 3; element 0 of a pair is the EVEN row
                const float s0 = odd ? G[0] : G[2], s1 = odd ? G[1] : G[3];
                const float q0 = acm_dpp<0xB1>(s0), q1 = acm_dpp<0xB1>(s1);       // quad_perm [1,0,3,2]
                const float x4[4] = {odd ? q0 : G[0], odd ? G[2] : q0, odd ? q1 : G[1], odd ? G[3] : q1};
                unsigned hh[2], mm[2], ll[2];
                split3_pairs(x4, hh, mm, ll);
                unsigned* dst = gsw + (4 * c + t) * 256;
                dst[0] = hh[0], dst[4] = hh[1];
                dst[AW_TILES * 256] = mm[0], dst[AW_TILES * 256 + 4] = mm[1];
                dst[2 * AW_TILES * 256] = ll[0], dst[2 * AW_TILES * 256 + 4] = ll[1];
            }
        }
            }
            aw_lds_barrier();                      // slab `it` is in its buffer; the other buffer is free again
        }
    // ---- head-parameter sums (as bwd_local16_kernel): value i = 4 t + r of lane (g, m = i) is column 16 t + 4 g + r
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
        // (the last barrier of the loop: every consumer is done with the operand buffers these slabs alias)
        float* slabv = lds + wv * AWB_NPG;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            slabv[(0 * 3 + c) * 64 + mycol] = dv[c];
            slabv[(1 * 3 + c) * 64 + mycol] = dgam[c];
            slabv[(2 * 3 + c) * 64 + mycol] = dbet[c];
        }
        if (lane < 9) {
            float v = dmix[0];
#pragma unroll
            for (int q = 1; q < 9; ++q) v = lane == q ? dmix[q] : v;
            slabv[9 * 64 + lane] = v;
        }
    } else {
    f32x4 W[2][3][4];                      // consumer wave j = wv - 4, tile ft: dW[c][feature 16 (j + 4 ft) + 4 g + r][column 16 t + (lane & 15)]
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) W[ft][c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // [feature tile][sub-slab][row]: a slab's operand rows are requested while the producers still work on it (one iteration
        // ahead of their use), all of them before the first product
        float pr[2][2][8], xr[2][2][8];
        auto request = [&](int slab_) {
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                if ((wv - 4) + 4 * ft >= KT) continue;
                const int feat = min(16 * ((wv - 4) + 4 * ft) + m, K - 1);
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const long rr = min((long)slab_ * AWB_SLAB + 32 * sb + 8 * g + e, (long)n_rows - 1);   // (G of such rows is 0)
                        pr[ft][sb][e] = agg[rr * ld_agg + feat];
                        xr[ft][sb][e] = xs[rr * ld_xs + feat];
                    }
            }
        };
        for (int it = 0; it <= my_slabs; ++it) {
            if (it > 0) {
        // ------------------------------------------------------------------ consumer: dW += A^T G for this wave's features
        const int slab = blockIdx.x + (it - 1) * gridDim.x;
        const u32x4* gbuf = Gs + ((it - 1) & 1) * (2 * 3 * AW_TILES * 64) + lane;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            if ((wv - 4) + 4 * ft >= KT) continue;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                float dr[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) dr[e] = xr[ft][sb][e] - pr[ft][sb][e];
                u32x4 ah[3], am[3], al3[3];
                split3(pr[ft][sb], ah[0], am[0], al3[0]);
                split3(dr, ah[1], am[1], al3[1]);
                split3(xr[ft][sb], ah[2], am[2], al3[2]);
                const u32x4* gb = gbuf + sb * 3 * AW_TILES * 64;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int t = 0; t < 4; t += 2) {                 // two column tiles at a time: their accumulators alternate
                        const u32x4 bh0 = gb[(0 * AW_TILES + 4 * c + t) * 64], bm0 = gb[(1 * AW_TILES + 4 * c + t) * 64],
                                    bl0 = gb[(2 * AW_TILES + 4 * c + t) * 64];
                        const u32x4 bh1 = gb[(0 * AW_TILES + 4 * c + t + 1) * 64], bm1 = gb[(1 * AW_TILES + 4 * c + t + 1) * 64],
                                    bl1 = gb[(2 * AW_TILES + 4 * c + t + 1) * 64];
                        f32x4 a0 = W[ft][c][t], a1 = W[ft][c][t + 1];
                        a0 = mma(al3[c], bh0, a0);
                        a1 = mma(al3[c], bh1, a1);
                        a0 = mma(ah[c], bl0, a0);
                        a1 = mma(ah[c], bl1, a1);
                        a0 = mma(am[c], bm0, a0);
                        a1 = mma(am[c], bm1, a1);
                        a0 = mma(am[c], bh0, a0);
                        a1 = mma(am[c], bh1, a1);
                        a0 = mma(ah[c], bm0, a0);
                        a1 = mma(ah[c], bm1, a1);
                        a0 = mma(ah[c], bh0, a0);
                        a1 = mma(ah[c], bh1, a1);
                        W[ft][c][t] = a0, W[ft][c][t + 1] = a1;
                    }
            }
        }
            }
            // (unconditional, from a clamped slab: a conditional request would keep the OLD rows alive through the whole iteration)
            request(min((int)(blockIdx.x + it * gridDim.x), nslab - 1));
            aw_lds_barrier();
        }
    // ---- the workgroup's weight-gradient partial, groups of 32 elements: part_w[group][block][32]
    if (wv >= 4) {
        const long gstride = (long)gridDim.x * 32;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            if ((wv - 4) + 4 * ft >= KT) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = c * (KT * 16 * 64) + (16 * ((wv - 4) + 4 * ft) + 4 * g + r) * 64 + 16 * t + m;
                        part_w[(long)(q >> 5) * gstride + blockIdx.x * 32 + (q & 31)] = W[ft][c][t][r];
                    }
        }
    }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < AWB_NPG; q += 512) {
        float sum = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 4; ++w8) sum += lds[w8 * AWB_NPG + q];            // (the four producer waves)
        part_head[(long)blockIdx.x * AWB_NPG + q] = sum;
    }
}

int aggw_bwd_blocks(int64_t n_rows) {
    int64_t nslab = (n_rows + AWB_SLAB - 1) / AWB_SLAB;
    return (int)(nslab < 256 ? (nslab < 1 ? 1 : nslab) : 256);      // one workgroup per CU (150 KB of LDS)
}
int aggw_feature_tiles(int64_t f_pad) { return (int)((f_pad + 15) / 16); }

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

extern "C" int acm_conv_aggw_bwd_workspace_bytes(int64_t n_rows, int64_t f_pad, size_t* bytes) {
    ACM_REQUIRE(bytes, ACM_EINVAL, "acm_conv_aggw_bwd_workspace_bytes: NULL argument");
    ACM_REQUIRE(n_rows >= 0 && f_pad > 0 && f_pad <= 128, ACM_ESHAPE, "acm_conv_aggw_bwd_workspace_bytes: bad shape");
    const int nblk = aggw_bwd_blocks(n_rows);
    *bytes = ((size_t)nblk * AWB_NPG + (size_t)nblk * 3 * aggw_feature_tiles(f_pad) * 16 * 64) * sizeof(float);
    return ACM_OK;
}

extern "C" int acm_conv_aggw_bwd(int64_t n_rows, int64_t f_in, int64_t f_pad, const float* agg, int64_t ld_agg, const float* xs,
                                 int64_t ld_xs, const acm_conv_bwd_local_t* p, float* d_w_low, float* d_w_high, float* d_w_mlp,
                                 int64_t ld_dw, void* workspace, size_t workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(p && agg && xs && d_w_low && d_w_high && d_w_mlp, ACM_EINVAL, "acm_conv_aggw_bwd: NULL argument");
    ACM_REQUIRE(n_rows >= 1 && n_rows < INT32_MAX - AWB_SLAB, ACM_ESHAPE, "acm_conv_aggw_bwd: bad row count");
    ACM_REQUIRE(p->f_out == 64 && p->n_channels == 3 && f_in > 0 && f_in <= f_pad && f_pad <= 128 && f_pad % 4 == 0 && ld_dw >= 64
                    && !p->post_scale && !p->g_scale,
                ACM_EUNSUPPORTED, "acm_conv_aggw_bwd: three fp32 channels of 64 columns, f_in <= f_pad <= 128, no post_scale / g_scale "
                "(got F %d, k %d, f_in %lld, f_pad %lld)", p->f_out, p->n_channels, (long long)f_in, (long long)f_pad);
    ACM_REQUIRE(p->grad_out && p->pre && p->s_mlp && p->att_mix && p->d_att_mix, ACM_EINVAL, "acm_conv_aggw_bwd: NULL tensor pointer");
    for (int c = 0; c < 3; ++c) {
        ACM_REQUIRE(p->att_vec[c] && p->d_att_vec[c], ACM_EINVAL, "acm_conv_aggw_bwd: att_vec[%d] NULL", c);
        ACM_REQUIRE(!p->layernorm || (p->ln_weight[c] && p->ln_bias[c] && p->d_ln_weight[c] && p->d_ln_bias[c]), ACM_EINVAL,
                    "acm_conv_aggw_bwd: layernorm pointers of channel %d NULL", c);
    }
    auto al16 = [](const void* q, int64_t ld) { return ((uintptr_t)q) % 16 == 0 && ld % 4 == 0; };
    ACM_REQUIRE(al16(p->pre, p->ld_pre) && al16(p->s_mlp, p->ld_s_mlp) && al16(p->grad_out, p->ld_grad_out), ACM_EUNSUPPORTED,
                "acm_conv_aggw_bwd: rows of pre / s_mlp / grad_out must be 16-byte aligned");
    size_t need = 0;
    acm_conv_aggw_bwd_workspace_bytes(n_rows, f_pad, &need);
    ACM_REQUIRE(workspace && workspace_bytes >= need, ACM_ENOMEM, "acm_conv_aggw_bwd: workspace %zu B < required %zu B",
                workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = aggw_bwd_blocks(n_rows), kt = aggw_feature_tiles(f_pad);
    float* part_w = (float*)workspace;                                  // (grouped slabs: 16-byte aligned lines)
    float* part_head = part_w + (size_t)nblk * 3 * kt * 16 * 64;
    const size_t lds = (size_t)AWB_GS * 16 + (576 + 192) * sizeof(float);
#define ACM_AGGWB(LNv, KTv)                                                                                                       \
    do {                                                                                                                          \
        ACM_CHECK_HIP(hipFuncSetAttribute((const void*)aggw_bwd_kernel<LNv, KTv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((aggw_bwd_kernel<LNv, KTv>), dim3(nblk), dim3(512), lds, st, *p, (int)n_rows, (int)f_pad, agg, (long)ld_agg, \
                           xs, (long)ld_xs, part_head, part_w);                                                                   \
    } while (0)
#define ACM_AGGWB_KT(LNv)                                                                                   \
    switch (kt) {                                                                                           \
        case 1: ACM_AGGWB(LNv, 1); break;  case 2: ACM_AGGWB(LNv, 2); break;  case 3: ACM_AGGWB(LNv, 3); break; \
        case 4: ACM_AGGWB(LNv, 4); break;  case 5: ACM_AGGWB(LNv, 5); break;  case 6: ACM_AGGWB(LNv, 6); break; \
        case 7: ACM_AGGWB(LNv, 7); break;  default: ACM_AGGWB(LNv, 8); break;                                \
    }
    if (p->layernorm) {
        ACM_AGGWB_KT(true)
    } else {
        ACM_AGGWB_KT(false)
    }
#undef ACM_AGGWB_KT
#undef ACM_AGGWB
    ACM_CHECK_HIP(hipGetLastError());
    // second phases: the head-parameter sums (the segments of acm_conv_bwd_local) and the three weight gradients (rows < f_in)
    acm_reduce_seg_t segs[13];
    int n = 0;
    for (int which = 0; which < 3; ++which)
        for (int c = 0; c < 3; ++c) {
            float* dst = which == 0 ? p->d_att_vec[c] : (which == 1 ? p->d_ln_weight[c] : p->d_ln_bias[c]);
            if (dst) segs[n++] = {part_head, nblk, AWB_NPG, (which * 3 + c) * 64, 64, dst, 64, 0, 0, 0, 0, 0};
        }
    segs[n++] = {part_head, nblk, AWB_NPG, 9 * 64, 9, p->d_att_mix, 9, 0, 0, 0, 0, 0};
    float* dws[3] = {d_w_low, d_w_high, d_w_mlp};
    for (int c = 0; c < 3; ++c)
        segs[n++] = {part_w, nblk, 32, c * kt * 16 * 64, (int32_t)(f_in * 64), dws[c], 64, 0, ld_dw, 0, (int32_t)((int64_t)nblk * 32), 0};
    return acm_reduce_emit(p->defer, segs, n, st);
}
