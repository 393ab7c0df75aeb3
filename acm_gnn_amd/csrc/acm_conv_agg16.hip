// Row-local stages of the aggregate-first ACM layer in the TRANSPOSED matrix-core layout (gfx950), for the
// reference's hidden width F = 64: three channels or four (structure_info: the fourth channel relu(deg (A_low S) - S)
// arrives as finished rows, ACM-Geometric/layers.py:110-113), f_pad = 4, 8 or 16 (ACM-Geometric/layers.py:57-63,101-108
// after P = A_low X has been gathered).  Template parameters: NC = channels, FP = f_pad (KB = FP / 4 contraction steps of
// v_mfma_f32_16x16x4_f32 per projection).
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


template <int NC, int FP, bool LN, bool NEXT>
__device__ __forceinline__ void epi16_body(const acm_conv_agg_fwd_t& p, int n_rows) {
    constexpr int KB = FP / 4;
    __shared__ __attribute__((aligned(16))) float ulds[NC * 64 + (NEXT ? ACM_E16_NEXT_LDS : 0)];
    const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    // u_c = gamma_c (.) att_vec_c (LayerNorm folded into the attention vector): with d = H - mean,
    //   s_c = sum_col (d * rstd * gamma + beta) * v = rstd * sum_col d * u_c + c0_c,   c0_c = sum_col beta_c * v_c
    // (compile-time channel indices only: a run-time index into the pointer arrays of the by-value argument struct -- also
    //  in the disguise of a four-way select chain, which LLVM folds back into one -- moves the whole struct to scratch)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if ((threadIdx.x >> 6) == (c & 3)) {
            float u = p.att_vec[c][lane];
            if (LN) u *= p.ln_weight[c][lane];
            ulds[c * 64 + lane] = u;
        }
    }
    if (NEXT) {                     // [col][8] = [W_L'(col, :) | W_H'(col, :) | W_I'(col, :) | 0]
        float* nlds = ulds + NC * 64;
        for (int idx = threadIdx.x; idx < ACM_E16_NEXT_LDS; idx += 256) {
            const int col = idx >> 3, j = idx & 7, c = j / p.next_f, q = j % p.next_f;
            const float* w = c == 0 ? p.next_w_low : (c == 1 ? p.next_w_high : p.next_w_mlp);
            nlds[idx] = (c < 3) ? w[(long)col * p.next_ld_w + q] : 0.f;
        }
    }
    float c0[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) c0[c] = LN ? acm_group_sum<64>(p.ln_bias[c][lane] * p.att_vec[c][lane]) : 0.f;
    // A operands: W_c[f = 4 kb + g][col = 16 t + m]
    float wreg[3][KB][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* w = c == 0 ? p.w_low : (c == 1 ? p.w_high : p.w_mlp);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                wreg[c][kb][t] = (4 * kb + g < p.f_in) ? w[(long)(4 * kb + g) * p.ld_w + 16 * t + m] : 0.f;
    }
    float mixm[NC * NC];
#pragma unroll
    for (int q = 0; q < NC * NC; ++q) mixm[q] = p.att_mix[q];
    __syncthreads();
    const AcmDropCtx dc = acm_drop_ctx(p.post_drop);
    AcmDropCtx ndc = dc;                           // (only read with next_x)
    if (p.next_x) ndc = acm_drop_ctx(p.next_drop);
    const float lo_a = p.relu_after ? 0.f : -INFINITY, lo_m = p.relu_mlp ? 0.f : -INFINITY;
    const float lo_post = p.post_relu ? 0.f : -INFINITY;
    const unsigned ld_agg = (unsigned)p.ld_agg, ld_xs = (unsigned)p.ld_xs, ld_out = (unsigned)p.ld_out;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;

    int base = wave * 16;
    if (base >= n_rows) return;
    // the operands of the NEXT step are requested before this step's math (one step of loads in flight)
    float nP[KB], nx[KB], nr[KB];                  // (nr: the raw input rows of the refill, next_x -- requested with the others)
    {
        const unsigned rr = (unsigned)min(base + m, n_rows - 1);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            nP[kb] = p.agg[rr * ld_agg + 4 * kb + g], nx[kb] = p.xs[rr * ld_xs + 4 * kb + g];
            nr[kb] = (p.next_x && 4 * kb + g < p.f_in) ? p.next_x[(long)rr * p.ld_next_x + 4 * kb + g] : 0.f;
        }
    }
    for (; base < n_rows; base += nwaves * 16) {
        const int row = base + m;
        const bool valid = row < n_rows;
        const unsigned rr = (unsigned)(valid ? row : n_rows - 1);
        float P[KB], x[KB], xraw[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) P[kb] = nP[kb], x[kb] = nx[kb], xraw[kb] = nr[kb];
        f32x4 D[NC][4];
        if (NC == 4) {                             // structure channel: relu(deg (A_low S) - S), finished rows of 64 floats
            const float dg = p.deg[rr];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 ps = *reinterpret_cast<const f32x4*>(p.ps + rr * (unsigned)p.ld_ps + 16 * t + 4 * g);
                const f32x4 ss = *reinterpret_cast<const f32x4*>(p.ss + rr * (unsigned)p.ld_ss + 16 * t + 4 * g);
                D[NC - 1][t] = dg * ps - ss;
            }
        }
        {
            const int nb = base + nwaves * 16;
            const unsigned r2 = (unsigned)min(nb + m, n_rows - 1);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                nP[kb] = p.agg[r2 * ld_agg + 4 * kb + g], nx[kb] = p.xs[r2 * ld_xs + 4 * kb + g];
                nr[kb] = (p.next_x && 4 * kb + g < p.f_in) ? p.next_x[(long)r2 * p.ld_next_x + 4 * kb + g] : 0.f;
            }
        }
        if (p.agg_copy && valid) {                 // the backward's operands (input pipeline): the rows just read
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                p.agg_copy[rr * (unsigned)p.ld_agg_copy + 4 * kb + g] = P[kb];
                p.xs_copy[rr * (unsigned)p.ld_xs_copy + 4 * kb + g] = x[kb];
            }
            if (p.next_x) {                        // ... and the NEXT step's dropped input over the elements just read (next_drop)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const int col = 4 * kb + g;
                    const_cast<float*>(p.xs)[rr * ld_xs + col] = col < p.f_in ? xraw[kb] * acm_drop1(ndc, row, col) : 0.f;
                }
            }
        }
        // (an opaque copy of the lane's group index: the LDS operands below depend on the lane only, and hoisted out of the
        //  row loop they would pin 48 .. 144 registers)
        const int gq = acm_opaque(g);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const float op[3] = {P[kb], x[kb] - P[kb], x[kb]};
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    D[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[c][kb][t], op[c], kb == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : D[c][t], 0, 0, 0);
        }
        // ---- head: statistics and attention scalars of row m (four lanes per row)
        float mean[NC], rstd[NC], gs[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float lo = c < 2 ? lo_a : (c == 2 ? lo_m : 0.f);
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
        float al[NC];
        {
            float lg[NC], mx = -INFINITY, den = 0.f;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) a = fmaf(gs[c], mixm[c * NC + j], a);
                lg[j] = a * (1.0f / NC);
                mx = fmaxf(mx, lg[j]);
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                lg[j] = acm_exp(lg[j] - mx);
                den += lg[j];
            }
            const float inv = acm_rcp(den);
#pragma unroll
            for (int j = 0; j < NC; ++j) al[j] = lg[j] * inv;
        }
        if (valid) {
            if (p.head_stats && g == 0) {          // [mean_c | rstd_c | sigmoid_c | alpha_c], NC each (row_head_store)
                float hv[4 * NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) hv[c] = mean[c], hv[NC + c] = rstd[c], hv[2 * NC + c] = gs[c], hv[3 * NC + c] = al[c];
                float* hs = p.head_stats + rr * (unsigned)p.ld_head_stats;
#pragma unroll
                for (int q = 0; q < NC; ++q) reinterpret_cast<float4*>(hs)[q] = make_float4(hv[4 * q], hv[4 * q + 1], hv[4 * q + 2], hv[4 * q + 3]);
            }
            if (g == 1) *reinterpret_cast<float4*>(p.att + (size_t)rr * 4) = make_float4(al[0], al[1], al[2], NC == 4 ? al[NC - 1] : 0.f);
        }
        // ---- mix, post-op, store; the row's next-layer projection
        const float a0 = al[0] * p.scale, a1 = al[1] * p.scale, a2 = al[2] * p.scale, a3 = al[NC - 1] * p.scale;
        float z8[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x4 o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaf(a2, D[2][t][r], fmaf(a1, D[1][t][r], a0 * D[0][t][r]));
                if (NC == 4) v = fmaf(a3, D[NC - 1][t][r], v);
                o[t][r] = fmaxf(v, lo_post);
            }
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
            const float* nlds = ulds + NC * 64;
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

template <int NC, int FP, bool LN, bool NEXT>
__global__ __launch_bounds__(256) void agg_epi16_kernel(acm_conv_agg_fwd_t p, int n_rows) {
    epi16_body<NC, FP, LN, NEXT>(p, n_rows);
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
// Structure channel (NC = 4): H_S = relu(deg (A_low S) - S) is read as finished rows (twice: for its dot with dO before the
// softmax backward, and again -- from the L1 / L2 -- for its own channel pass, so that its 16 registers are not held
// across the three projected channels); its G_S leaves as 16-byte stores (deg * G_S for an explicit operator) for the
// F-wide transposed product d struc_low = A_low^T (D G_S) - G_S that follows (acm_spmm_ex); no dW for it.
#define ACM_B16_TS 68                  /* floats per tile row: 16-byte writes of eight consecutive row-lanes and the B-operand reads
                                          (rows 4 g + s: two row groups per LDS pass, 16 banks apart) are conflict-free */

template <int NC, int FP, bool PROJ>
struct B16Lds {
    static constexpr int KB = FP / 4;
    static constexpr int PS = 2 * FP + 1;                                   // floats per [P | x] row (odd: no bank conflicts)
    static constexpr int TILES = 4 * 16 * ACM_B16_TS, PX = 4 * 16 * PS + 16 - (4 * 16 * PS) % 16;   // (16-byte aligned end)
    static constexpr int HL = 3 * NC * 64, UL = NC * 64, WL = 3 * KB * 4 * 64, PW = PROJ ? 512 : 0;
    static constexpr int FRONT = TILES + PX + HL + UL + WL + PW;
    static constexpr int NPG0 = 3 * FP * 64 + 3 * NC * 64 + NC * NC;         // f_in = FP at most
    static constexpr int SLAB = ((NPG0 + 31) & ~31) + (PROJ ? 6 * 64 : 0);
    static constexpr int TOTAL = (FRONT > 4 * SLAB ? FRONT : 4 * SLAB) + 64;
    static_assert(TOTAL * 4 <= 64 * 1024, "static LDS of one workgroup");
};

// PROJ: the following layer's projection backward rides along (acm_conv_agg_bwd_t.proj_*): grad_out is formed per row from
// proj_dz (6 floats) and the 64 x 6 weight table in LDS instead of being read (256 B per row), and proj_d_w = out^T proj_dz
// is one more set of MFMAs over the `out` tile (which passes through the same LDS tile as the G_c do).
// GATHER: a workgroup of EIGHT waves, one per SIMD for each role: waves 0-3 run this backward, waves 4-7 walk the operator's
// id streams for the next training step's P = A_low dropout(x) (acm_conv_agg_bwd_t.next_agg; stream_gather_role).  The
// kernel is compiled for two waves per SIMD either way, so the pair costs the backward no occupancy; its waves get the
// vector and matrix pipes almost to themselves (the gather waves wait on memory), the gather waves the memory system.
template <int NC, int FP, bool LN, bool OUT_MASK, bool PROJ, bool GATHER>
__device__ __forceinline__ void bwd16_body(const acm_conv_agg_bwd_t& p, int n_rows, float* __restrict__ partial, const GatherRole* gr) {
    using L = B16Lds<NC, FP, PROJ>;
    constexpr int KB = L::KB, PS = L::PS;
    __shared__ __attribute__((aligned(16))) float lds[L::TOTAL];
    const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    float* gt = lds + wv * (16 * ACM_B16_TS);                 // this wave's G tile
    float* px = lds + L::TILES + wv * 16 * PS;                // this wave's [P | x] rows, 2 FP floats each
    float* hl = lds + L::TILES + L::PX;                       // [att_vec | gamma | beta][c][col]; 16-byte aligned
    float* ul = hl + L::HL;                                   // u_c = att_vec_c * gamma_c
    float* wl = ul + L::UL;                                   // A operands of the projections, [(c, kb, t)][lane]
    float* pw = wl + L::WL;                                   // PROJ: [col][8] = [W_L'(col, :) | W_H'(col, :) | W_I'(col, :) | 0]
    const int f_in = p.f_in;
    // one round of independent global loads, one barrier: the head parameters, u = att_vec * gamma, the projections' A
    // operands W_c[f = 4 kb + g][col = 16 t + m] (the same for every wave), c1_c = mean_col(u_c)
    // (compile-time channel indices only: a run-time index into the pointer arrays of the by-value argument struct -- also
    //  in the disguise of a four-way select chain, which LLVM folds back into one -- moves the whole struct to scratch and
    //  turns every load of the kernel into a flat load: 142 us instead of ~80 for the four-channel kernel)
    float c1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float av = p.att_vec[c][lane];
        const float gw = LN ? p.ln_weight[c][lane] : 1.f, gb = LN ? p.ln_bias[c][lane] : 0.f;
        const float u = av * gw;
        if (wv == (c & 3)) {
            hl[c * 64 + lane] = av, hl[NC * 64 + c * 64 + lane] = gw, hl[2 * NC * 64 + c * 64 + lane] = gb;
            ul[c * 64 + lane] = u;
        }
        // (wave-uniform: kept in a scalar register -- readfirstlane is an integer builtin, hence the bit casts)
        c1[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acm_group_sum<64>(u) * (1.0f / 64.0f))));
    }
    for (int idx = threadIdx.x; idx < 3 * KB * 4 * 64; idx += blockDim.x) {
        const int e = idx >> 6, l2 = idx & 63, c = e / (4 * KB), kb = (e >> 2) % KB, t = e & 3, f = 4 * kb + (l2 >> 4);
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
    float mixm[NC * NC];
#pragma unroll
    for (int q = 0; q < NC * NC; ++q) mixm[q] = p.att_mix[q];
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
    const float fsel = m < FP ? 1.f : 0.f;          // A operand of the dW products: feature f = m (< f_pad)

    f32x4 acc[3][4], acc2[4];            // acc2 (PROJ): proj_d_w^T tiles, D[i = j of proj_dz][col]
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned ld_pz = (unsigned)p.ld_proj_dz;
    // pA[c]: lane (g, m) accumulates column 16 (m >> 2) + 4 g + (m & 3) of A_c; dmx[j]: lane group g accumulates row c = g of
    // d att_mix for its own row m (NC accumulators per lane instead of NC^2; summed over the 16 row-lanes after the loop)
    constexpr int NDM = NC == 3 ? 9 : NC;                // (three channels: the nine accumulators of round 3, registers permitting)
    float pA[NC], pS[NC], dmx[NDM];
#pragma unroll
    for (int c = 0; c < NC; ++c) pS[c] = pA[c] = 0.f;
#pragma unroll
    for (int q = 0; q < NDM; ++q) dmx[q] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int base = wave * 16;
    float nP[KB], nx[KB];
    f32x4 ngo[4], nou[4], nst[NC];
    // the operands of the projections (P, x: 8 FP bytes per row) are requested one step ahead; grad_out / out (512 B per row),
    // the head statistics and proj_dz at the top of their own step -- the projections' MFMAs run while they arrive
#define ACM_B16_LOAD(BASE)                                                                              \
    do {                                                                                                \
        const unsigned r2 = (unsigned)min((BASE) + m, n_rows - 1);                                      \
        _Pragma("unroll") for (int kb = 0; kb < KB; ++kb)                                               \
            nP[kb] = p.agg[r2 * ld_agg + 4 * kb + g], nx[kb] = p.xs[r2 * ld_xs + 4 * kb + g];           \
    } while (0)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) nP[kb] = nx[kb] = 0.f;
    if (base < n_rows) ACM_B16_LOAD(base);
    for (; base < n_rows; base += nwaves * 16) {
        const bool valid = base + m < n_rows;
        float P[KB], x[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) P[kb] = nP[kb], x[kb] = nx[kb];
        float dzr[6], dza[4];            // PROJ: proj_dz of row m; of rows 4 g + s at column m (the A operand of proj_d_w)
        const unsigned r1 = (unsigned)min(base + m, n_rows - 1);
        float dgs = 0.f;                 // NC == 4: deg of row m
        {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!PROJ) ngo[t] = *reinterpret_cast<const f32x4*>(p.grad_out + r1 * ld_go + 16 * t + 4 * g);
                if (out_mask || PROJ) nou[t] = *reinterpret_cast<const f32x4*>(p.out + r1 * ld_out + 16 * t + 4 * g);
            }
#pragma unroll
            for (int q = 0; q < NC; ++q) nst[q] = *reinterpret_cast<const f32x4*>(p.head_stats + r1 * ld_hs + 4 * q);
            if (NC == 4) dgs = p.deg[r1];
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
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) px[mq * PS + 4 * kb + gq] = P[kb], px[mq * PS + FP + 4 * kb + gq] = x[kb];
        // RECOMP (four channels): a projected channel lives in 16 registers at a time -- computed for its dot product with dO,
        // dropped, and computed AGAIN (8 MFMAs: the matrix pipe is a quarter busy) in its own pass below -- instead of all
        // three (48 registers) across both passes: with the structure channel's extra state the kernel otherwise spills
        constexpr bool RECOMP = NC == 4;
        f32x4 D[RECOMP ? 1 : 3][4];
        const int lq = acm_opaque(lane);
#define ACM_B16_PROJECT_C(CH, DST)                                                                                      \
        _Pragma("unroll") for (int kb = 0; kb < KB; ++kb) {                                                             \
            const float opc = (CH) == 0 ? P[kb] : ((CH) == 1 ? x[kb] - P[kb] : x[kb]);                                  \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                               \
                DST[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[(((CH) * KB + kb) * 4 + t) * 64 + lq], opc,            \
                                                              kb == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : DST[t], 0, 0, 0); \
        }
        if (!RECOMP) {                               // three channels: all of them now, the MFMAs run while grad_out arrives
#pragma unroll                                       // (twelve independent MFMAs per contraction step)
            for (int kb = 0; kb < KB; ++kb) {
                const float op[3] = {P[kb], x[kb] - P[kb], x[kb]};
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        D[RECOMP ? 0 : c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[((c * KB + kb) * 4 + t) * 64 + lq], op[c],
                                                                                kb == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : D[RECOMP ? 0 : c][t], 0, 0, 0);
            }
        }
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
        float mean[NC], rstd[NC], gsig[NC], al[NC];      // head_stats: [mean_c | rstd_c | sigmoid_c | alpha_c], NC each
        {
            float hv[4 * NC];
#pragma unroll
            for (int q = 0; q < NC; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[4 * q + r] = nst[q][r];
#pragma unroll
            for (int c = 0; c < NC; ++c) mean[c] = hv[c], rstd[c] = hv[NC + c], gsig[c] = hv[2 * NC + c], al[c] = hv[3 * NC + c];
        }
        // ---- mix / softmax / sigmoid backward: ds_c = dL/ds_c per row
        float dal[NC], ds[NC];
        if (NC == 4) {                               // H_S dot dO (the rows are read again in the channel's own pass)
            float part = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 ps = *reinterpret_cast<const f32x4*>(p.ps + r1 * (unsigned)p.ld_ps + 16 * t + 4 * g);
                const f32x4 ss = *reinterpret_cast<const f32x4*>(p.ss + r1 * (unsigned)p.ld_ss + 16 * t + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fmaf(dO[t][r], fmaxf(fmaf(dgs, ps[r], -ss[r]), 0.f), part);
            }
            dal[NC - 1] = p.scale * row4_sum(part);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float lo = c < 2 ? lo_a : lo_m;
            float part = 0.f;
            if (RECOMP) ACM_B16_PROJECT_C(c, D[0])
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    D[RECOMP ? 0 : c][t][r] = fmaxf(D[RECOMP ? 0 : c][t][r], lo);
                    part = fmaf(dO[t][r], D[RECOMP ? 0 : c][t][r], part);
                }
            dal[c] = p.scale * row4_sum(part);
        }
        {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < NC; ++j) dot = fmaf(al[j], dal[j], dot);
            float dlg[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) dlg[j] = al[j] * (dal[j] - dot);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float dg = 0.f;
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    dg = fmaf(dlg[j], mixm[c * NC + j], dg);
                    if (NC == 3) dmx[(c * NC + j) % NDM] = fmaf(wq * gsig[c], dlg[j] * (1.0f / NC), dmx[(c * NC + j) % NDM]);
                }
                ds[c] = dg * (1.0f / NC) * gsig[c] * (1.f - gsig[c]);
                pS[c] = fmaf(wq, ds[c], pS[c]);
            }
            if (NC == 4) {
                const float gsel = valid ? (g == 0 ? gsig[0] : (g == 1 ? gsig[1] : (g == 2 ? gsig[2] : gsig[NC - 1]))) : 0.f;
#pragma unroll
                for (int j = 0; j < NC; ++j) dmx[j % NDM] = fmaf(gsel, dlg[j] * (1.0f / NC), dmx[j % NDM]);
            }
        }
        // ---- one channel at a time: G_c -> LDS tile -> dW_c on the matrix pipe (the structure channel: G_S -> memory)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float lo = c < 2 ? lo_a : (c == 2 ? lo_m : 0.f);
            const float aal = p.scale * al[c];
            f32x4 HS[4];                               // c == 3: the channel's rows, again (an opaque row index: the compiler
            if (c == 3) {                              // must not keep the first read's 16 registers alive instead)
                const unsigned r3 = (unsigned)acm_opaque((int)r1);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 ps = *reinterpret_cast<const f32x4*>(p.ps + r3 * (unsigned)p.ld_ps + 16 * t + 4 * g);
                    const f32x4 ss = *reinterpret_cast<const f32x4*>(p.ss + r3 * (unsigned)p.ld_ss + 16 * t + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) HS[t][r] = fmaxf(fmaf(dgs, ps[r], -ss[r]), 0.f);
                }
            }
            if (RECOMP && c < 3) {                     // the channel again (same MFMA chain: bit-identical to the first time)
                ACM_B16_PROJECT_C(c, D[0])
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) D[0][t][r] = fmaxf(D[0][t][r], lo);
            }
#define ACM_B16_H(t, r) (c == 3 ? HS[t][r] : D[(RECOMP || c > 2) ? 0 : c][t][r])
            // two passes over the lane's 16 columns, four at a time, so that neither xhat nor G is held as a whole: (A) the
            // row sums (contrib -> the reduce-scatter; t2 = sum_col u xhat), (B) xhat again, G, straight into the LDS tile
            float contrib[16], t2 = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(ul + c * 64 + 16 * t + 4 * gq);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = LN ? (ACM_B16_H(t, r) - mean[c]) * rstd[c] : ACM_B16_H(t, r);
                    contrib[4 * t + r] = ds[c] * xh;
                    if (LN) t2 = fmaf(u[r], xh, t2);
                }
            }
            pA[c] += row_reduce_scatter16(contrib, mq);
            const float m1 = LN ? ds[c] * c1[c] : 0.f, m2 = LN ? ds[c] * row4_sum(t2) * (1.0f / 64.0f) : 0.f;
            const float dg1 = (c == 3 && p.g_struc_scale) ? p.g_struc_scale[r1] : 1.f;
            const int gq2 = NC == 4 ? acm_opaque(gq) : gq;   // (four channels: u is read again, not kept across the reduce-scatter)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(ul + c * 64 + 16 * t + 4 * gq2);
                f32x4 G;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v;
                    if (LN) {
                        const float xh = (ACM_B16_H(t, r) - mean[c]) * rstd[c];
                        v = fmaf(aal, dO[t][r], rstd[c] * (fmaf(ds[c], u[r], -m1) - xh * m2));
                    } else {
                        v = fmaf(aal, dO[t][r], ds[c] * u[r]);
                    }
                    G[r] = ACM_B16_H(t, r) > lo ? v : 0.f;
                }
                if (c == 3) {
                    if (valid) *reinterpret_cast<f32x4*>(p.g_struc + r1 * (unsigned)p.ld_g_struc + 16 * t + 4 * g) = dg1 * G;
                } else {
                    *reinterpret_cast<f32x4*>(gt + mq * ACM_B16_TS + 16 * t + 4 * gq) = G;
                }
            }
#undef ACM_B16_H
            if (c < 3) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    // A operand: feature m of row 4 g + s of this wave step (P | x rows parked in LDS at the top of the step)
                    const float ap = px[(4 * gq + s) * PS + (mq & (FP - 1))], ax = px[(4 * gq + s) * PS + FP + (mq & (FP - 1))];
                    const float aop = fsel * (c == 0 ? ap : (c == 1 ? ax - ap : ax));
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[c < 3 ? c : 0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, gt[(4 * gq + s) * ACM_B16_TS + 16 * t + mq], acc[c < 3 ? c : 0][t], 0, 0, 0);
                }
            }
        }
    }
#undef ACM_B16_PROJECT_C
#undef ACM_B16_LOAD
    // ---- end of the row loop: head-parameter sums over the 16 row-lanes, then the block's partial slab
    const int npg0 = 3 * f_in * 64 + 3 * NC * 64 + NC * NC;
    const int off2 = (npg0 + 31) & ~31;              // PROJ: proj_d_w behind d_params at a whole group of 32 (the second phase
    const int npg = PROJ ? off2 + 64 * nq : npg0;    // sums it by lines), [c][col][q] = flat index (j / f') 64 f' + col f' + j % f'
    // value i = 4 t + r of lane (g, m = i) is column 16 t + 4 g + r: one column of A_c per lane
    const int mycol = 16 * (m >> 2) + 4 * g + (m & 3);
    float dv[NC], dgam[NC], dbet[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        pS[c] = acm_group_sum<64>(pS[c]);
        const float v = hl[c * 64 + mycol], gm = hl[NC * 64 + c * 64 + mycol], bt = hl[2 * NC * 64 + c * 64 + mycol];
        dv[c] = fmaf(gm, pA[c], bt * pS[c]);
        dgam[c] = v * pA[c];
        dbet[c] = v * pS[c];
    }
#pragma unroll
    for (int q = 0; q < NDM; ++q) dmx[q] = NC == 3 ? acm_group_sum<64>(dmx[q]) : acm_group_sum<16>(dmx[q]);   // (four channels: the
                                                                   // 16 rows of the lane group: d att_mix[g][j])
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
        for (int c = 0; c < NC; ++c) {
            slab[b2 + (0 * NC + c) * 64 + mycol] = dv[c];
            slab[b2 + (1 * NC + c) * 64 + mycol] = dgam[c];
            slab[b2 + (2 * NC + c) * 64 + mycol] = dbet[c];
        }
    }
    if (NC == 3) {
        if (lane < 9) {
            float v = dmx[0];
#pragma unroll
            for (int q = 1; q < 9; ++q) v = lane == q ? dmx[q % NDM] : v;
            slab[3 * f_in * 64 + 9 * 64 + lane] = v;
        }
    } else if (m == 0) {
#pragma unroll
        for (int j = 0; j < NC; ++j) slab[3 * f_in * 64 + 3 * NC * 64 + g * NC + j] = dmx[j % NDM];
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

template <int NC, int FP, bool LN, bool OUT_MASK, bool PROJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void agg_bwd16_kernel(acm_conv_agg_bwd_t p, int n_rows, float* __restrict__ partial) {
    bwd16_body<NC, FP, LN, OUT_MASK, PROJ, false>(p, n_rows, partial, nullptr);
}
template <int NC, bool LN, bool PROJ>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void agg_bwd16_gather_kernel(acm_conv_agg_bwd_t p, int n_rows, float* __restrict__ partial,
                                                                                                   GatherRole gr) {
    bwd16_body<NC, 8, LN, true, PROJ, true>(p, n_rows, partial, &gr);
}

}  // namespace

// The row-local forward stage over an existing P = A_low X (p->agg; with four channels also p->ps = A_low S).  Returns
// ACM_OK after a launch, -1 when the configuration is not one this kernel is written for (the caller then runs
// agg_epilogue_kernel), or an error.
int acm_agg_epi16(const acm_conv_agg_fwd_t* p, int64_t n_rows, bool* next_done, hipStream_t s) {
    *next_done = false;
    const int NC = p->n_channels, FP = p->f_pad;
    if ((NC != 3 && NC != 4) || (FP != 4 && FP != 8 && FP != 16) || p->f_out != 64 || !(acm_tuning().rows16 & ACM_ROWS16_EPI)) return -1;
    int64_t ld_max = 64;
    for (int64_t ld : {p->ld_agg, p->ld_xs, p->ld_out, p->ld_head_stats, p->ld_post_scale, p->ld_agg_copy, p->ld_xs_copy, p->ld_ps, p->ld_ss})
        ld_max = ld > ld_max ? ld : ld_max;
    if (n_rows * ld_max >= (int64_t)INT32_MAX) return -1;                  // 32-bit element offsets
    if ((((uintptr_t)p->out) % 16) != 0 || (p->ld_out % 4) != 0) return -1;
    if (p->post_scale && ((((uintptr_t)p->post_scale) % 16) != 0 || (p->ld_post_scale % 4) != 0)) return -1;
    if (NC == 4 && ((((uintptr_t)p->ps) % 16) != 0 || (((uintptr_t)p->ss) % 16) != 0 || p->ld_ps % 4 != 0 || p->ld_ss % 4 != 0)) return -1;
    const bool next = p->next_f > 0;
    int grid = (int)((n_rows + 63) / 64);
    if (grid > 1024) grid = 1024;                    // four workgroups (sixteen waves) per CU
#define ACM_E16_L(NCv, FPv)                                                                                                   \
    do {                                                                                                                     \
        if (p->layernorm) {                                                                                                  \
            if (next) hipLaunchKernelGGL((agg_epi16_kernel<NCv, FPv, true, true>), dim3(grid), dim3(256), 0, s, *p, (int)n_rows);   \
            else hipLaunchKernelGGL((agg_epi16_kernel<NCv, FPv, true, false>), dim3(grid), dim3(256), 0, s, *p, (int)n_rows);       \
        } else {                                                                                                             \
            if (next) hipLaunchKernelGGL((agg_epi16_kernel<NCv, FPv, false, true>), dim3(grid), dim3(256), 0, s, *p, (int)n_rows);  \
            else hipLaunchKernelGGL((agg_epi16_kernel<NCv, FPv, false, false>), dim3(grid), dim3(256), 0, s, *p, (int)n_rows);      \
        }                                                                                                                    \
    } while (0)
#define ACM_E16(NCv)                            \
    do {                                        \
        if (FP == 4) ACM_E16_L(NCv, 4);         \
        else if (FP == 8) ACM_E16_L(NCv, 8);    \
        else ACM_E16_L(NCv, 16);                \
    } while (0)
    if (NC == 3) ACM_E16(3);
    else ACM_E16(4);
#undef ACM_E16
#undef ACM_E16_L
    ACM_CHECK_HIP(hipGetLastError());
    *next_done = next;
    return ACM_OK;
}

// Floats of the per-block partial slab the sixteen-rows-per-wave backward writes (acm_conv_agg_bwd_workspace_bytes).
// The row-local backward over the forward's head_stats.  Returns the number of blocks launched (> 0), 0 when the
// configuration is not this kernel's (the caller runs agg_bwd_kernel), or a negative acm_status_t.  The partial slab has
// npg + 64 * 3 * proj_f entries per block (acm_conv_agg_bwd_t.proj_*: the following layer's weight gradient behind d_params).
int acm_agg_bwd16(const acm_conv_agg_bwd_t* p, int64_t n_rows, float* partial, int max_blocks, hipStream_t s, const GatherRole* gr,
                  int gather_blocks) {
    const int NC = p->n_channels, FP = p->f_pad;
    if ((NC != 3 && NC != 4) || (FP != 4 && FP != 8 && FP != 16) || p->f_out != 64 || !p->head_stats ||
        !(acm_tuning().rows16 & ACM_ROWS16_BWD))
        return 0;
    const bool out_mask = p->out != nullptr && p->post_relu && !p->post_scale;
    const bool no_post = !p->post_relu && !p->post_scale && !(p->post_drop.p > 0.f);
    if (!out_mask && !no_post) return 0;
    const bool proj = p->proj_dz != nullptr;
    if ((proj || gr) && (FP != 8 || !out_mask)) return 0;      // the carriers exist for the hidden layer of a training step
    if (proj && (!p->out || p->proj_f < 1 || p->proj_f > 2 || !p->proj_w_low || !p->proj_w_high || !p->proj_w_mlp || !p->proj_d_w ||
                 p->ld_proj_dz < 3 * p->proj_f || p->proj_ld_w < p->proj_f || n_rows * p->ld_proj_dz >= (int64_t)INT32_MAX))
        return 0;
    for (const void* q : {(const void*)(proj ? nullptr : p->grad_out), (const void*)p->out, (const void*)p->head_stats,
                          (const void*)(NC == 4 ? p->ps : nullptr), (const void*)(NC == 4 ? p->ss : nullptr),
                          (const void*)(NC == 4 ? p->g_struc : nullptr)})
        if (((uintptr_t)q) % 16 != 0) return 0;
    if ((!proj && p->ld_grad_out % 4 != 0) || ((out_mask || proj) && p->ld_out % 4 != 0) || p->ld_head_stats % 4 != 0) return 0;
    if (NC == 4 && (p->ld_ps % 4 != 0 || p->ld_ss % 4 != 0 || p->ld_g_struc % 4 != 0 || !p->deg)) return 0;
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
    const bool ln = p->layernorm != 0;
#define ACM_B16_BASE(NCv, FPv)                                                                                                     \
    do {                                                                                                                          \
        if (ln) {                                                                                                                 \
            if (out_mask) hipLaunchKernelGGL((agg_bwd16_kernel<NCv, FPv, true, true, false>), dim3(grid), dim3(256), 0, s, q, (int)n_rows, partial);   \
            else hipLaunchKernelGGL((agg_bwd16_kernel<NCv, FPv, true, false, false>), dim3(grid), dim3(256), 0, s, q, (int)n_rows, partial);           \
        } else {                                                                                                                  \
            if (out_mask) hipLaunchKernelGGL((agg_bwd16_kernel<NCv, FPv, false, true, false>), dim3(grid), dim3(256), 0, s, q, (int)n_rows, partial);  \
            else hipLaunchKernelGGL((agg_bwd16_kernel<NCv, FPv, false, false, false>), dim3(grid), dim3(256), 0, s, q, (int)n_rows, partial);          \
        }                                                                                                                         \
    } while (0)
    // the carriers (FP = 8, the output read behind a fused ReLU): + proj_*, + next_agg, or both
#define ACM_B16_CARRY(NCv)                                                                                                          \
    do {                                                                                                                          \
        if (gr) {                                                                                                                 \
            if (ln) { if (proj) hipLaunchKernelGGL((agg_bwd16_gather_kernel<NCv, true, true>), dim3(grid), dim3(512), 0, s, q, (int)n_rows, partial, *gr);   \
                      else hipLaunchKernelGGL((agg_bwd16_gather_kernel<NCv, true, false>), dim3(grid), dim3(512), 0, s, q, (int)n_rows, partial, *gr); }       \
            else { if (proj) hipLaunchKernelGGL((agg_bwd16_gather_kernel<NCv, false, true>), dim3(grid), dim3(512), 0, s, q, (int)n_rows, partial, *gr);     \
                   else hipLaunchKernelGGL((agg_bwd16_gather_kernel<NCv, false, false>), dim3(grid), dim3(512), 0, s, q, (int)n_rows, partial, *gr); }         \
        } else {                                                                                                                  \
            if (ln) hipLaunchKernelGGL((agg_bwd16_kernel<NCv, 8, true, true, true>), dim3(grid), dim3(256), 0, s, q, (int)n_rows, partial);                   \
            else hipLaunchKernelGGL((agg_bwd16_kernel<NCv, 8, false, true, true>), dim3(grid), dim3(256), 0, s, q, (int)n_rows, partial);                     \
        }                                                                                                                         \
    } while (0)
    if (proj || gr) {
        if (NC == 3) ACM_B16_CARRY(3);
        else ACM_B16_CARRY(4);
    } else if (NC == 3) {
        if (FP == 4) ACM_B16_BASE(3, 4);
        else if (FP == 8) ACM_B16_BASE(3, 8);
        else ACM_B16_BASE(3, 16);
    } else {
        if (FP == 4) ACM_B16_BASE(4, 4);
        else if (FP == 8) ACM_B16_BASE(4, 8);
        else ACM_B16_BASE(4, 16);
    }
#undef ACM_B16_BASE
#undef ACM_B16_CARRY
    if (hipGetLastError() != hipSuccess) return -ACM_EHIP;
    return grid;
}
