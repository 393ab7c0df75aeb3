// Device-side building blocks shared by the ACM layer kernels (acm_conv.hip, acm_conv_agg.hip):
// lane layouts, the adaptive-mixing head (LayerNorm -> att_vec dot -> sigmoid -> k x k mix ->
// softmax) and the parameter-gradient accumulators of its backward.
#pragma once
#include <math.h>

#include "acm_common.h"

// ------------------------------------------------------------------ layouts
// How the F columns of one row are spread over lanes.  NV = values per lane.
template <int NREG>
struct LayWide {  // A: wave per row, lane owns columns lane + 64 i
    static constexpr int NV = NREG;
    int lane;
    __device__ __forceinline__ int col(int i) const { return lane + 64 * i; }
    __device__ __forceinline__ float rsum(float v) const { return acm_group_sum<64>(v); }
    __device__ __forceinline__ bool leader() const { return lane == 0; }
};
template <int FP>
struct LayPacked {  // B: FP lanes per row (64 / FP rows per wave), one column per lane
    static constexpr int NV = 1;
    int lane;
    __device__ __forceinline__ int col(int) const { return lane % FP; }
    __device__ __forceinline__ float rsum(float v) const { return acm_group_sum<FP>(v); }
    __device__ __forceinline__ bool leader() const { return (lane % FP) == 0; }
};
template <int NVv>
struct LayGrouped {  // D: 16 lanes (= one DPP row) per matrix row, 4 rows per wave; lane m owns
                     //    columns m, m+16, ... (NV of them) -- the B-operand layout of mfma 16x16x4
    static constexpr int NV = NVv;
    int lane;
    __device__ __forceinline__ int col(int i) const { return (lane & 15) + 16 * i; }
    __device__ __forceinline__ float rsum(float v) const { return acm_group_sum<16>(v); }
    __device__ __forceinline__ bool leader() const { return (lane & 15) == 0; }
};
struct LayPair32 {  // E: 32 lanes per row, lane l owns the adjacent columns 2l, 2l+1 (one packed bf16 pair);
                    //    both half-waves hold the same row after the half-combine, the lower one stores
    static constexpr int NV = 2;
    int lane;
    __device__ __forceinline__ int col(int i) const { return 2 * (lane & 31) + i; }
    __device__ __forceinline__ float rsum(float v) const { return acm_group_sum<32>(v); }
    __device__ __forceinline__ bool leader() const { return (lane & 31) == 0; }
};
template <int NB>
struct LayVec16 {  // F: 16 lanes x float4 per 64-column block (NB blocks): lane m owns columns 64 b + 4 m .. + 3 -- the
                   //    layout a dwordx4 gather of row segments leaves behind (spmm_vec_kernel).  After the cross-group
                   //    sum all four 16-lane groups of the wave hold the same row; group 0 stores
    static constexpr int NV = 4 * NB;
    int lane;
    __device__ __forceinline__ int col(int i) const { return 64 * (i >> 2) + 4 * (lane & 15) + (i & 3); }
    __device__ __forceinline__ float rsum(float v) const { return acm_group_sum<16>(v); }
    __device__ __forceinline__ bool leader() const { return lane == 0; }
};
struct LayRow16 {   // G: the columns of layout F (one 64-column block), but each 16-lane group holds ITS OWN row -- four rows per
                    //    wave, every group stores: the row-local head as a kernel of its own (conv_head_rows_kernel)
    static constexpr int NV = 4;
    int lane;
    __device__ __forceinline__ int col(int i) const { return 4 * (lane & 15) + (i & 3); }
    __device__ __forceinline__ float rsum(float v) const { return acm_group_sum<16>(v); }
    __device__ __forceinline__ bool leader() const { return (lane & 15) == 0; }
};
template <int FP>
struct LaySerial {  // C: every lane holds the whole row
    static constexpr int NV = FP;
    bool lead;
    __device__ __forceinline__ int col(int i) const { return i; }
    __device__ __forceinline__ float rsum(float v) const { return v; }
    __device__ __forceinline__ bool leader() const { return lead; }
};

// 1-ulp hardware transcendentals for the mixing head.  The IEEE sequences hipcc emits for
// expf (~20 instructions) and a / b (~10) made the row-local kernels VALU-bound; v_exp_f32 /
// v_rcp_f32 / v_rsq_f32 are single instructions.  acm_exp keeps the exponent argument exact to
// ~2^-48 (product error and the low word of log2 e folded back in), so the result stays within
// ~2 ulp for |x| < 80 instead of drifting with |x| * 2^-24.
__device__ __forceinline__ float acm_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float acm_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float acm_exp(float x) {
    const float L_HI = 1.44269502162933349609375f;   // float(log2 e)
    const float L_LO = 1.92596299112661746e-8f;      // log2 e - L_HI
    const float t = x * L_HI;
    float lo = fmaf(x, L_HI, -t);
    lo = fmaf(x, L_LO, lo);
    return __builtin_amdgcn_exp2f(t) * fmaf(lo, 0.693147182464599609375f, 1.0f);
}

struct GatherSrc {
    const float* p[3];
    long ld[3];
};

// ------------------------------------------------------------------ attention head (shared by fwd / bwd)
struct HeadOut {
    float g[4], alpha[4], rstd[4];
};
struct HeadParams {  // copied out of the kernel-argument struct so every index is a constant
    const float* att_vec[4];
    const float* ln_w[4];
    const float* ln_b[4];
    const float* att_mix;
};
template <class P>
__device__ __forceinline__ HeadParams acm_head_params(const P& p) {
    HeadParams h;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        h.att_vec[c] = p.att_vec[c];
        h.ln_w[c] = p.ln_weight[c];
        h.ln_b[c] = p.ln_bias[c];
    }
    h.att_mix = p.att_mix;
    return h;
}

// H: activated channels; hn/xhat outputs (hn = LayerNorm(H) or H).  Invalid columns hold 0.
template <class L, int K>
__device__ __forceinline__ void acm_head(const L& lay, int F, int layernorm,
                                         const HeadParams& hp,
                                         const float (&H)[4][L::NV], float (&hn)[4][L::NV],
                                         float (&xhat)[4][L::NV], HeadOut& o) {
    constexpr int NV = L::NV;
    constexpr int k = K;
    const float invF = 1.0f / (float)F;
    float s[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= K) {
            o.g[c] = 0.f;
            o.rstd[c] = 1.f;
            continue;
        }
        // Without LayerNorm the same straight-line code runs with mean 0, rstd 1, gamma 1,
        // beta 0 (exact: (H - 0) * 1 * 1 + 0 == H), so only two scalars depend on the branch.
        float mean = 0.f, rstd = 1.f;
        if (layernorm) {
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) part += H[c][i];
            mean = lay.rsum(part) * invF;
            part = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float d = (lay.col(i) < F) ? (H[c][i] - mean) : 0.f;
                part += d * d;
            }
            rstd = acm_rsqrt(lay.rsum(part) * invF + ACM_LN_EPS);
        }
        o.rstd[c] = rstd;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = lay.col(i);
            const bool ok = col < F;
            const float gam = (ok && layernorm) ? hp.ln_w[c][col] : 1.f;
            const float bet = (ok && layernorm) ? hp.ln_b[c][col] : 0.f;
            const float xh = ok ? (H[c][i] - mean) * rstd : 0.f;
            xhat[c][i] = xh;
            hn[c][i] = ok ? (xh * gam + bet) : 0.f;
        }
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = lay.col(i);
            part += (col < F) ? hn[c][i] * hp.att_vec[c][col] : 0.f;
        }
        s[c] = lay.rsum(part);
        o.g[c] = acm_rcp(1.0f + acm_exp(-s[c]));
    }
    float logit[4], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j >= k) {
            logit[j] = -INFINITY;
            continue;
        }
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < k) acc += o.g[c] * hp.att_mix[c * k + j];
        logit[j] = acc * (1.0f / (float)k);
        mx = fmaxf(mx, logit[j]);
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j >= k) continue;
        logit[j] = acm_exp(logit[j] - mx);
        den += logit[j];
    }
    const float inv_den = acm_rcp(den);
#pragma unroll
    for (int j = 0; j < 4; ++j) o.alpha[j] = (j < k) ? logit[j] * inv_den : 0.f;
}

// Parameter-gradient vector layout of the head (npg = 3 k F + k k floats):
//   [ d att_vec : k x F ][ d ln_weight : k x F ][ d ln_bias : k x F ][ d att_mix : k x k ]
template <class L>
struct ParamAcc {
    float dv[4][L::NV], dgam[4][L::NV], dbet[4][L::NV], dmix[16];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < L::NV; ++i) dv[c][i] = dgam[c][i] = dbet[c][i] = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) dmix[q] = 0.f;
    }
};


// Backward of acm_head + the mix  out = scale * sum_c alpha_c H_c :
//   in : dO (grad of out), the forward quantities (H, hn, xhat, ho)
//   out: dH[c] = dL/dH_c (before any ReLU mask); parameter gradients accumulated into `pa`,
//        weighted by `act` (0 for padding rows).
template <class L, int K>
__device__ __forceinline__ void acm_head_backward(const L& lay, int F, int layernorm, const HeadParams& hp,
                                                  float scale, const float (&H)[4][L::NV],
                                                  const float (&hn)[4][L::NV], const float (&xhat)[4][L::NV],
                                                  const HeadOut& ho, const float (&dO)[L::NV], float act,
                                                  ParamAcc<L>& pa, float (&dH)[4][L::NV]) {
    constexpr int NV = L::NV;
    constexpr int k = K;
    float dalpha[4], dot = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= k) {
            dalpha[c] = 0.f;
            continue;
        }
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) part += dO[i] * H[c][i];
        dalpha[c] = scale * lay.rsum(part);
        dot += ho.alpha[c] * dalpha[c];
    }
    float dlogit[4], ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dlogit[j] = (j < k) ? ho.alpha[j] * (dalpha[j] - dot) : 0.f;
    const float invk = 1.0f / (float)k;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= k) {
            ds[c] = 0.f;
            continue;
        }
        float dgc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < k) {
                dgc += dlogit[j] * hp.att_mix[c * k + j];
                pa.dmix[c * 4 + j] += act * ho.g[c] * dlogit[j] * invk;
            }
        dgc *= invk;
        ds[c] = dgc * ho.g[c] * (1.f - ho.g[c]);
    }
    const float invF = 1.0f / (float)F;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= k) {
#pragma unroll
            for (int i = 0; i < NV; ++i) dH[c][i] = 0.f;
            continue;
        }
        if (layernorm) {
            float dxh[NV], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int col = lay.col(i);
                const bool ok = col < F;
                const float v = ok ? hp.att_vec[c][col] : 0.f;
                const float gam = ok ? hp.ln_w[c][col] : 0.f;
                const float dhn = ds[c] * v;
                pa.dgam[c][i] += act * dhn * xhat[c][i];
                pa.dbet[c][i] += act * dhn;
                pa.dv[c][i] += act * ds[c] * hn[c][i];
                dxh[i] = dhn * gam;
                s1 += dxh[i];
                s2 += dxh[i] * xhat[c][i];
            }
            const float m1 = lay.rsum(s1) * invF, m2 = lay.rsum(s2) * invF;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                dH[c][i] = scale * ho.alpha[c] * dO[i] + ho.rstd[c] * (dxh[i] - m1 - xhat[c][i] * m2);
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int col = lay.col(i);
                const float v = (col < F) ? hp.att_vec[c][col] : 0.f;
                pa.dv[c][i] += act * ds[c] * hn[c][i];
                dH[c][i] = scale * ho.alpha[c] * dO[i] + ds[c] * v;
            }
        }
    }
}

// =====================================================================================
// Lean row-local head for the 4-rows-per-wave layout (LayGrouped<4>: lane m of a 16-lane group owns
// columns m, m+16, m+32, m+48; F <= 64).  Register discipline: the head parameters live in LDS,
// pass 1 keeps only scalars, pass 2 handles one channel at a time.
// =====================================================================================

// att_vec / LayerNorm gamma, beta staged in LDS as [array][c][m][i] (array 0 = att_vec, 1 = gamma,
// 2 = beta; without LayerNorm gamma = 1, beta = 0; zero beyond F): one ds_read_b128 per use, and --
// unlike loads from global memory -- nothing for the compiler to hoist out of the row loop.
template <int K>
__device__ __forceinline__ void stage_head_params(float* hlds, const float* const* att_vec, const float* const* ln_w,
                                                  const float* const* ln_b, int layernorm, int F) {
    for (int idx = threadIdx.x; idx < 3 * K * 64; idx += 256) {
        const int arr = idx / (K * 64), c = (idx / 64) % K, m = (idx % 64) / 4, i = idx % 4;
        const int col = m + 16 * i;
        float v = 0.f;
        if (col < F) {
            if (arr == 0) v = att_vec[c][col];
            else if (layernorm) v = (arr == 1) ? ln_w[c][col] : ln_b[c][col];
            else v = (arr == 1) ? 1.f : 0.f;
        }
        hlds[idx] = v;
    }
}

// Make a loop-invariant lane index opaque to the optimiser.  The head parameters in LDS are indexed
// by the lane only, so LICM hoists every ds_read_b128 of them out of the row loop and pins 36-48 VGPRs
// for the whole kernel; re-reading 9-12 x 16 B from LDS per row is far cheaper than a lost wave.
__device__ __forceinline__ int acm_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <int K>
struct RowHead {
    float mean[K], rstd[K], gsig[K], alpha[K];
};

// head_stats row (acm_conv_agg_fwd_t): mean[K] | rstd[K] | gsig[K] | alpha[K], 16-byte aligned
template <int K>
__device__ __forceinline__ void row_head_store(float* __restrict__ dst, const RowHead<K>& r) {
    float v[4 * K];
#pragma unroll
    for (int c = 0; c < K; ++c) v[c] = r.mean[c], v[K + c] = r.rstd[c], v[2 * K + c] = r.gsig[c], v[3 * K + c] = r.alpha[c];
#pragma unroll
    for (int q = 0; q < K; ++q)
        reinterpret_cast<float4*>(dst)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
template <int K>
__device__ __forceinline__ void row_head_load(const float* __restrict__ src, RowHead<K>& r) {
    float v[4 * K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const float4 t = reinterpret_cast<const float4*>(src)[q];
        v[4 * q] = t.x, v[4 * q + 1] = t.y, v[4 * q + 2] = t.z, v[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int c = 0; c < K; ++c) r.mean[c] = v[c], r.rstd[c] = v[K + c], r.gsig[c] = v[2 * K + c], r.alpha[c] = v[3 * K + c];
}

template <int K>
struct HeadVecs {   // the lane's four columns of att_vec / gamma / beta of one channel
    float v[4], gm[4], bt[4];
    __device__ __forceinline__ void load(const float* hlds, int c, int m) {
        const float4 a = *reinterpret_cast<const float4*>(hlds + ((0 * K + c) * 16 + m) * 4);
        const float4 b = *reinterpret_cast<const float4*>(hlds + ((1 * K + c) * 16 + m) * 4);
        const float4 d = *reinterpret_cast<const float4*>(hlds + ((2 * K + c) * 16 + m) * 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        gm[0] = b.x; gm[1] = b.y; gm[2] = b.z; gm[3] = b.w;
        bt[0] = d.x; bt[1] = d.y; bt[2] = d.z; bt[3] = d.w;
    }
};

// Pass 1: channel statistics and attention scalars from the activated channels H (mixm = K x K mix, SGPRs).
template <int K>
__device__ __forceinline__ void row_head(const float* hlds, const float* mixm, int m, int F, bool ln,
                                         const float (&H)[K][4], RowHead<K>& r) {
    const float invF = 1.0f / (float)F;
#pragma unroll
    for (int c = 0; c < K; ++c) {
        HeadVecs<K> hv;
        hv.load(hlds, c, m);
        float mu = 0.f, rs = 1.f;
        if (ln) {
            mu = acm_group_sum<16>((H[c][0] + H[c][1]) + (H[c][2] + H[c][3])) * invF;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = (m + 16 * i < F) ? H[c][i] - mu : 0.f;
                q = fmaf(d, d, q);
            }
            rs = acm_rsqrt(acm_group_sum<16>(q) * invF + ACM_LN_EPS);
        }
        r.mean[c] = mu;
        r.rstd[c] = rs;
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float hn = (m + 16 * i < F) ? fmaf((H[c][i] - mu) * rs, hv.gm[i], hv.bt[i]) : 0.f;
            dot = fmaf(hn, hv.v[i], dot);
        }
        r.gsig[c] = acm_rcp(1.0f + acm_exp(-acm_group_sum<16>(dot)));
    }
    float lg[K], mx = -INFINITY, den = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < K; ++c) a = fmaf(r.gsig[c], mixm[c * K + j], a);
        lg[j] = a * (1.0f / (float)K);
        mx = fmaxf(mx, lg[j]);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        lg[j] = acm_exp(lg[j] - mx);
        den += lg[j];
    }
    const float inv = acm_rcp(den);
#pragma unroll
    for (int j = 0; j < K; ++j) r.alpha[j] = lg[j] * inv;
}

// Undo the forward's fused post-op on the incoming gradient (raw = out before the post-op).
template <int K, class P>
__device__ __forceinline__ void row_post_backward(const P& p, const RowHead<K>& r, const float (&H)[K][4], bool active,
                                                  long row, int m, int F, float (&dO)[4]) {
    const bool drop = p.post_drop.p > 0.f;
    if (!(p.post_relu || p.post_scale || drop)) return;
    float df[4];
    acm_drop4(acm_drop_ctx(p.post_drop), row, m, df);      // the forward's mask, regenerated
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float raw = 0.f;
#pragma unroll
        for (int c = 0; c < K; ++c) raw = fmaf(r.alpha[c], H[c][i], raw);
        if (p.post_relu && !(raw * p.scale > 0.f)) dO[i] = 0.f;
        if (p.post_scale && active && m + 16 * i < F) dO[i] *= p.post_scale[row * p.ld_post_scale + m + 16 * i];
        if (drop) dO[i] *= df[i];
    }
}

// Backward through mix / softmax / sigmoid: ds[c] = dL/ds_c; accumulates d att_mix (K x K).
template <int K>
__device__ __forceinline__ void row_head_backward_scalars(const RowHead<K>& r, const float* mixm, float scale,
                                                          const float (&H)[K][4], const float (&dO)[4],
                                                          float (&ds)[K], int qc, int qj, float& dmix1) {
    float dal[K], dot = 0.f;
#pragma unroll
    for (int c = 0; c < K; ++c) {
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) part = fmaf(dO[i], H[c][i], part);
        dal[c] = scale * acm_group_sum<16>(part);
        dot = fmaf(r.alpha[c], dal[c], dot);
    }
    float dlg[K];
#pragma unroll
    for (int j = 0; j < K; ++j) dlg[j] = r.alpha[j] * (dal[j] - dot);
    const float invk = 1.0f / (float)K;
#pragma unroll
    for (int c = 0; c < K; ++c) {
        float dg = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) dg = fmaf(dlg[j], mixm[c * K + j], dg);
        ds[c] = dg * invk * r.gsig[c] * (1.f - r.gsig[c]);
    }
    // d att_mix[c][j] += gsig[c] * dlg[j] / k: the K x K terms are uniform in the 16-lane group, so lane m keeps
    // element m = (qc, qj) only (one accumulator per lane instead of K * K); rows with dO = 0 add 0
    float gq = r.gsig[0], lq = dlg[0];
#pragma unroll
    for (int c = 1; c < K; ++c) {
        gq = (qc == c) ? r.gsig[c] : gq;
        lq = (qj == c) ? dlg[c] : lq;
    }
    dmix1 = fmaf(gq, lq * invk, dmix1);
}

// Pass 2 for one channel: G = dL/dH_c (before the ReLU mask), accumulating d att_vec / d gamma / d beta.
template <int K>
__device__ __forceinline__ void row_channel_backward(const float* hlds, int c, int m, int F, bool ln, float scale,
                                                     const RowHead<K>& r, float ds_c, const float (&Hc)[4],
                                                     const float (&dO)[4], float (&A)[4], float& S, float (&G)[4]) {
    // Parameter gradients of the head are rank-structured in the per-row scalar ds_c = dL/ds_c:
    //     d att_vec = gamma * A + beta * S,   d gamma = att_vec * A,   d beta = att_vec * S
    // with A[col] = sum_rows ds_c * xhat[row][col] and S = sum_rows ds_c, so a channel needs 4 + 1 accumulators per
    // lane instead of 12 (row_param_grads() expands them after the row loop).  Without LayerNorm xhat = H, gamma = 1,
    // beta = 0 (what stage_head_params stores), and the same formulas give d att_vec = A.
    HeadVecs<K> hv;
    hv.load(hlds, c, m);
    S += ds_c;
    if (ln) {
        const float invF = 1.0f / (float)F;
        float xh[4], dxh[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = m + 16 * i < F;
            xh[i] = ok ? (Hc[i] - r.mean[c]) * r.rstd[c] : 0.f;
            A[i] = fmaf(ds_c, xh[i], A[i]);
            dxh[i] = ds_c * hv.v[i] * hv.gm[i];
            s1 += dxh[i];
            s2 = fmaf(dxh[i], xh[i], s2);
        }
        const float m1 = acm_group_sum<16>(s1) * invF, m2 = acm_group_sum<16>(s2) * invF;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            G[i] = fmaf(scale * r.alpha[c], dO[i], r.rstd[c] * (dxh[i] - m1 - xh[i] * m2));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            A[i] = fmaf(ds_c, Hc[i], A[i]);
            G[i] = fmaf(scale * r.alpha[c], dO[i], ds_c * hv.v[i]);
        }
    }
}

// (A, S) summed over all rows -> the lane's four columns of d att_vec / d gamma / d beta of channel c
template <int K>
__device__ __forceinline__ void row_param_grads(const float* hlds, int c, int m, const float (&A)[4], float S,
                                                float (&dv)[4], float (&dgam)[4], float (&dbet)[4]) {
    HeadVecs<K> hv;
    hv.load(hlds, c, m);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dv[i] = fmaf(hv.gm[i], A[i], hv.bt[i] * S);
        dgam[i] = hv.v[i] * A[i];
        dbet[i] = hv.v[i] * S;
    }
}
