// Aggregate-first form of the ACM layer (K2a forward, K3a backward) for gfx950.
//
//   P = A_low X  (narrow gather: F_in <= 16 floats per edge instead of 2 F)       -- lanes over neighbours
//   pre_L = P W_L, pre_H = (X - P) W_H, Z_I = X W_I  (3 F_in FMAs per column)      -- lanes over columns
//   ReLU / LayerNorm / sigmoid / 3x3 mix / softmax / weighted sum                  -- shared head
//
// One wave per work item; waves are persistent (grid-stride over items) so the three
// F_in x F weight panels stay in registers (lane l holds column l of every row of W).
// The backward needs no transposed SpMM when the layer input has no gradient:
//   dW_L = P^T G_L, dW_H = (X - P)^T G_H, dW_I = X^T G_I
// are outer-product reductions over rows, accumulated per lane in registers and combined
// deterministically (wave -> LDS -> per-block partial -> tree reduce).
#include "acm_conv_device.h"

namespace {

template <int FP>
struct AggWeights {  // lane l: column l of W_low / W_high / W_mlp, rows 0..FP-1 (zero beyond f_in)
    float wl[FP], wh[FP], wm[FP];
    __device__ __forceinline__ void load(const float* w_low, const float* w_high, const float* w_mlp, long ld,
                                         int f_in, int F, int lane) {
#pragma unroll
        for (int f = 0; f < FP; ++f) {
            const bool ok = f < f_in && lane < F;
            wl[f] = ok ? w_low[(long)f * ld + lane] : 0.f;
            wh[f] = ok ? w_high[(long)f * ld + lane] : 0.f;
            wm[f] = ok ? w_mlp[(long)f * ld + lane] : 0.f;
        }
    }
};

template <int FP>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&v)[FP]) {
#pragma unroll
    for (int q = 0; q < FP / 4; ++q) {
        const float4 t = reinterpret_cast<const float4*>(p)[q];
        v[4 * q + 0] = t.x;
        v[4 * q + 1] = t.y;
        v[4 * q + 2] = t.z;
        v[4 * q + 3] = t.w;
    }
}

// P (wave-uniform) -> projections -> head -> out / att / agg for one row.
template <int FP>
__device__ __forceinline__ void agg_fwd_row(const acm_conv_agg_fwd_t& p, const AggWeights<FP>& W, int row,
                                            int lane, const float (&P)[FP]) {
    using L = LayWide<1>;
    const int F = p.f_out;
    float x[FP];
    load_vec<FP>(p.xs + (long)row * p.ld_xs, x);   // same address in every lane: one broadcast fetch
    float p0 = 0.f, p1 = 0.f, zi = 0.f;
#pragma unroll
    for (int f = 0; f < FP; ++f) {
        p0 = fmaf(P[f], W.wl[f], p0);
        p1 = fmaf(x[f] - P[f], W.wh[f], p1);
        zi = fmaf(x[f], W.wm[f], zi);
    }
    float H[4][1], hn[4][1], xhat[4][1];
    const bool ok = lane < F;
    H[0][0] = ok ? (p.relu_after ? fmaxf(p0, 0.f) : p0) : 0.f;
    H[1][0] = ok ? (p.relu_after ? fmaxf(p1, 0.f) : p1) : 0.f;
    H[2][0] = ok ? (p.relu_mlp ? fmaxf(zi, 0.f) : zi) : 0.f;
    H[3][0] = 0.f;
    L lay{lane};
    HeadOut ho;
    const HeadParams hp = acm_head_params(p);
    acm_head<L, 3>(lay, F, p.layernorm, hp, H, hn, xhat, ho);
    if (ok)
        p.out[(long)row * p.ld_out + lane] =
            p.scale * (ho.alpha[0] * H[0][0] + ho.alpha[1] * H[1][0] + ho.alpha[2] * H[2][0]);
    if (lane == 0) {
        *reinterpret_cast<float4*>(p.att + (long)row * 4) = make_float4(ho.alpha[0], ho.alpha[1], ho.alpha[2], 0.f);
        float* ag = p.agg + (long)row * p.ld_agg;
#pragma unroll
        for (int q = 0; q < FP / 4; ++q)
            reinterpret_cast<float4*>(ag)[q] = make_float4(P[4 * q], P[4 * q + 1], P[4 * q + 2], P[4 * q + 3]);
    }
}

template <int FP>
__global__ __launch_bounds__(256) void agg_fwd_kernel(CsrView csr, acm_conv_agg_fwd_t p, float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    AggWeights<FP> W;
    W.load(p.w_low, p.w_high, p.w_mlp, p.ld_w, p.f_in, p.f_out, lane);
    for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < csr.n_items; w += nw) {
        const AcmItem it = csr.items[w];
        const int row = acm_uniform(it.row), begin = acm_uniform(it.begin), end = acm_uniform(it.end),
                  slot = acm_uniform(it.slot);
        float acc[FP];
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[f] = 0.f;
        for (int k0 = begin; k0 < end; k0 += 128) {      // 2 neighbours per lane in flight
            const int ka = k0 + lane, kb = ka + 64;
            const bool va = ka < end, vb = kb < end;
            const int ja = va ? csr.indices[ka] : 0, jb = vb ? csr.indices[kb] : 0;
            const float aa = va ? csr.vals[ka] : 0.f, ab = vb ? csr.vals[kb] : 0.f;
            float xa[FP], xb[FP];
            load_vec<FP>(p.xg + (long)ja * p.ld_xg, xa);
            load_vec<FP>(p.xg + (long)jb * p.ld_xg, xb);
#pragma unroll
            for (int f = 0; f < FP; ++f) {
                acc[f] = va ? fmaf(aa, xa[f], acc[f]) : acc[f];
                acc[f] = vb ? fmaf(ab, xb[f], acc[f]) : acc[f];
            }
        }
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[f] = acm_group_sum<64>(acc[f]);
        if (slot < 0) {
            agg_fwd_row<FP>(p, W, row, lane, acc);
        } else if (lane == 0) {
            float* ps = partial + (long)slot * FP;
#pragma unroll
            for (int q = 0; q < FP / 4; ++q)
                reinterpret_cast<float4*>(ps)[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
    }
}

template <int FP>
__global__ __launch_bounds__(256) void agg_fwd_fixup_kernel(CsrView csr, acm_conv_agg_fwd_t p,
                                                            const float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    AggWeights<FP> W;
    W.load(p.w_low, p.w_high, p.w_mlp, p.ld_w, p.f_in, p.f_out, lane);
    for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < csr.n_long; w += nw) {
        const AcmLongRow lr = csr.long_rows[w];
        const int row = acm_uniform(lr.row), sb = acm_uniform(lr.slot_begin), se = acm_uniform(lr.slot_end);
        // lanes split the slots, then a fixed-order butterfly combines them (deterministic)
        float acc[FP];
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[f] = 0.f;
        for (int s = sb + lane; s < se; s += 64) {
            float v[FP];
            load_vec<FP>(partial + (long)s * FP, v);
#pragma unroll
            for (int f = 0; f < FP; ++f) acc[f] += v[f];
        }
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[f] = acm_group_sum<64>(acc[f]);
        agg_fwd_row<FP>(p, W, row, lane, acc);
    }
}

// ---------------------------------------------------------------- backward
// flat parameter-gradient vector: [dW_low f_in*F][dW_high][dW_mlp][dv 3F][dgamma 3F][dbeta 3F][dmix 9]
template <int FP>
__global__ __launch_bounds__(256) void agg_bwd_kernel(acm_conv_agg_bwd_t p, int n_rows, float* __restrict__ partial) {
    extern __shared__ float lds[];
    using L = LayWide<1>;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int F = p.f_out, f_in = p.f_in;
    const int npg = 3 * f_in * F + 9 * F + 9;
    AggWeights<FP> W;
    W.load(p.w_low, p.w_high, p.w_mlp, p.ld_w, f_in, F, lane);
    float dwl[FP], dwh[FP], dwm[FP];
#pragma unroll
    for (int f = 0; f < FP; ++f) dwl[f] = dwh[f] = dwm[f] = 0.f;
    ParamAcc<L> pa;
    pa.zero();
    L lay{lane};
    const HeadParams hp = acm_head_params(p);
    const bool ok = lane < F;
    for (int row = blockIdx.x * 4 + wv; row < n_rows; row += gridDim.x * 4) {
        float P[FP], x[FP];
        load_vec<FP>(p.agg + (long)row * p.ld_agg, P);
        load_vec<FP>(p.xs + (long)row * p.ld_xs, x);
        float dO[1];
        dO[0] = ok ? p.grad_out[(long)row * p.ld_grad_out + lane] : 0.f;
        float p0 = 0.f, p1 = 0.f, zi = 0.f;
#pragma unroll
        for (int f = 0; f < FP; ++f) {
            p0 = fmaf(P[f], W.wl[f], p0);
            p1 = fmaf(x[f] - P[f], W.wh[f], p1);
            zi = fmaf(x[f], W.wm[f], zi);
        }
        const bool pos0 = p.relu_after ? (p0 > 0.f) : true, pos1 = p.relu_after ? (p1 > 0.f) : true,
                   pos2 = p.relu_mlp ? (zi > 0.f) : true;
        float H[4][1], hn[4][1], xhat[4][1], dH[4][1];
        H[0][0] = (ok && pos0) ? p0 : 0.f;
        H[1][0] = (ok && pos1) ? p1 : 0.f;
        H[2][0] = (ok && pos2) ? zi : 0.f;
        H[3][0] = 0.f;
        HeadOut ho;
        acm_head<L, 3>(lay, F, p.layernorm, hp, H, hn, xhat, ho);
        acm_head_backward<L, 3>(lay, F, p.layernorm, hp, p.scale, H, hn, xhat, ho, dO, 1.f, pa, dH);
        const float g0 = (ok && pos0) ? dH[0][0] : 0.f, g1 = (ok && pos1) ? dH[1][0] : 0.f,
                    g2 = (ok && pos2) ? dH[2][0] : 0.f;
#pragma unroll
        for (int f = 0; f < FP; ++f) {
            dwl[f] = fmaf(P[f], g0, dwl[f]);
            dwh[f] = fmaf(x[f] - P[f], g1, dwh[f]);
            dwm[f] = fmaf(x[f], g2, dwm[f]);
        }
    }
    float* slab = lds + wv * npg;
    if (ok) {
#pragma unroll
        for (int f = 0; f < FP; ++f)
            if (f < f_in) {
                slab[(0 * f_in + f) * F + lane] = dwl[f];
                slab[(1 * f_in + f) * F + lane] = dwh[f];
                slab[(2 * f_in + f) * F + lane] = dwm[f];
            }
        const int base = 3 * f_in * F;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            slab[base + (0 * 3 + c) * F + lane] = pa.dv[c][0];
            slab[base + (1 * 3 + c) * F + lane] = pa.dgam[c][0];
            slab[base + (2 * 3 + c) * F + lane] = pa.dbet[c][0];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 3; ++j) slab[3 * f_in * F + 9 * F + c * 3 + j] = pa.dmix[c * 4 + j];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < npg; q += 256)
        partial[(long)blockIdx.x * npg + q] = (lds[q] + lds[npg + q]) + (lds[2 * npg + q] + lds[3 * npg + q]);
}

// dst[q] = sum_b partial[b][q], fixed tree order; grid = npg blocks.
__global__ __launch_bounds__(256) void reduce_columns_kernel(const float* __restrict__ partial, int nblk, int npg,
                                                             float* __restrict__ dst) {
    __shared__ float red[256];
    const int q = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) s += partial[(long)b * npg + q];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) dst[q] = red[0];
}

int agg_pad(int f_in) { return f_in <= 4 ? 4 : (f_in <= 8 ? 8 : 16); }

int agg_bwd_blocks(int64_t n_rows) {
    int64_t nb = (n_rows + 15) / 16;
    if (nb > 768) nb = 768;
    if (nb < 1) nb = 1;
    return (int)nb;
}

template <class P>
int check_common(const P* p, const char* who) {
    ACM_REQUIRE(p->f_in >= 1 && p->f_in <= 16, ACM_EUNSUPPORTED, "%s: f_in %d outside 1..16", who, p->f_in);
    ACM_REQUIRE(p->f_out >= 1 && p->f_out <= 64, ACM_EUNSUPPORTED, "%s: f_out %d outside 1..64", who, p->f_out);
    ACM_REQUIRE(p->f_pad == agg_pad(p->f_in), ACM_ESHAPE, "%s: f_pad %d, expected %d for f_in %d", who, p->f_pad,
                agg_pad(p->f_in), p->f_in);
    ACM_REQUIRE(p->w_low && p->w_high && p->w_mlp && p->att_mix && p->xs, ACM_EINVAL, "%s: NULL pointer", who);
    ACM_REQUIRE(p->ld_w >= p->f_out, ACM_ESHAPE, "%s: ld_w too small", who);
    ACM_REQUIRE(((uintptr_t)p->xs) % 16 == 0 && (p->ld_xs * 4) % 16 == 0 && p->ld_xs >= p->f_pad, ACM_EINVAL,
                "%s: xs rows must be 16-byte aligned and f_pad long", who);
    for (int c = 0; c < 3; ++c) {
        ACM_REQUIRE(p->att_vec[c], ACM_EINVAL, "%s: att_vec[%d] NULL", who, c);
        ACM_REQUIRE(!p->layernorm || (p->ln_weight[c] && p->ln_bias[c]), ACM_EINVAL, "%s: LayerNorm pointers NULL", who);
    }
    return ACM_OK;
}

}  // namespace

extern "C" int acm_conv_agg_fwd(const acm_csr_t* a, const acm_conv_agg_fwd_t* p, void* workspace,
                                size_t workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(a && p, ACM_EINVAL, "acm_conv_agg_fwd: NULL argument");
    int st = check_common(p, "acm_conv_agg_fwd");
    if (st != ACM_OK) return st;
    ACM_REQUIRE(p->xg && p->out && p->agg && p->att, ACM_EINVAL, "acm_conv_agg_fwd: NULL tensor pointer");
    ACM_REQUIRE(((uintptr_t)p->xg) % 16 == 0 && (p->ld_xg * 4) % 16 == 0 && p->ld_xg >= p->f_pad &&
                    ((uintptr_t)p->agg) % 16 == 0 && (p->ld_agg * 4) % 16 == 0 && p->ld_agg >= p->f_pad &&
                    ((uintptr_t)p->att) % 16 == 0, ACM_EINVAL,
                "acm_conv_agg_fwd: xg / agg rows must be 16-byte aligned and f_pad long");
    const size_t need = (size_t)a->n_slots * p->f_pad * sizeof(float);
    ACM_REQUIRE(workspace_bytes >= need && (need == 0 || workspace), ACM_ENOMEM,
                "acm_conv_agg_fwd: workspace %zu B < required %zu B", workspace_bytes, need);
    if (a->n_items == 0) return ACM_OK;
    hipStream_t s = (hipStream_t)stream;
    const CsrView v = acm_view(a);
    float* partial = (float*)workspace;
    int grid = (int)((a->n_items + 3) / 4);
    if (grid > 2048) grid = 2048;
    int gridf = (int)((a->n_long + 3) / 4);
    if (gridf > 1024) gridf = 1024;
#define ACM_AGG(FPv)                                                                                        \
    do {                                                                                                    \
        hipLaunchKernelGGL((agg_fwd_kernel<FPv>), dim3(grid), dim3(256), 0, s, v, *p, partial);             \
        if (a->n_long)                                                                                      \
            hipLaunchKernelGGL((agg_fwd_fixup_kernel<FPv>), dim3(gridf), dim3(256), 0, s, v, *p, partial);  \
    } while (0)
    if (p->f_pad == 4) ACM_AGG(4);
    else if (p->f_pad == 8) ACM_AGG(8);
    else ACM_AGG(16);
#undef ACM_AGG
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}

extern "C" int acm_conv_agg_bwd_workspace_bytes(int64_t n_rows, int f_in, int f_out, size_t* bytes) {
    ACM_REQUIRE(bytes, ACM_EINVAL, "acm_conv_agg_bwd_workspace_bytes: NULL argument");
    ACM_REQUIRE(f_in >= 1 && f_in <= 16 && f_out >= 1 && f_out <= 64, ACM_EUNSUPPORTED,
                "acm_conv_agg_bwd_workspace_bytes: f_in %d f_out %d unsupported", f_in, f_out);
    const size_t npg = (size_t)3 * f_in * f_out + 9 * (size_t)f_out + 9;
    *bytes = (size_t)agg_bwd_blocks(n_rows) * npg * sizeof(float);
    return ACM_OK;
}

extern "C" int acm_conv_agg_bwd(int64_t n_rows, const acm_conv_agg_bwd_t* p, void* workspace,
                                size_t workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(p, ACM_EINVAL, "acm_conv_agg_bwd: NULL argument");
    int st = check_common(p, "acm_conv_agg_bwd");
    if (st != ACM_OK) return st;
    ACM_REQUIRE(p->grad_out && p->agg && p->d_params, ACM_EINVAL, "acm_conv_agg_bwd: NULL tensor pointer");
    ACM_REQUIRE(((uintptr_t)p->agg) % 16 == 0 && (p->ld_agg * 4) % 16 == 0 && p->ld_agg >= p->f_pad, ACM_EINVAL,
                "acm_conv_agg_bwd: agg rows must be 16-byte aligned and f_pad long");
    size_t need = 0;
    acm_conv_agg_bwd_workspace_bytes(n_rows, p->f_in, p->f_out, &need);
    ACM_REQUIRE(workspace && workspace_bytes >= need, ACM_ENOMEM, "acm_conv_agg_bwd: workspace %zu B < required %zu B",
                workspace_bytes, need);
    const int npg = 3 * p->f_in * p->f_out + 9 * p->f_out + 9;
    const int nblk = agg_bwd_blocks(n_rows);
    const size_t lds = (size_t)4 * npg * sizeof(float);
    ACM_REQUIRE(lds <= 64 * 1024, ACM_EUNSUPPORTED, "acm_conv_agg_bwd: %zu B of LDS needed", lds);
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    if (p->f_pad == 4)
        hipLaunchKernelGGL((agg_bwd_kernel<4>), dim3(nblk), dim3(256), lds, s, *p, (int)n_rows, partial);
    else if (p->f_pad == 8)
        hipLaunchKernelGGL((agg_bwd_kernel<8>), dim3(nblk), dim3(256), lds, s, *p, (int)n_rows, partial);
    else
        hipLaunchKernelGGL((agg_bwd_kernel<16>), dim3(nblk), dim3(256), lds, s, *p, (int)n_rows, partial);
    ACM_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(reduce_columns_kernel, dim3(npg), dim3(256), 0, s, partial, nblk, npg, p->d_params);
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}
