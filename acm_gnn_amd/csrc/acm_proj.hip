// Backward of a skinny projection Z = X W (W: f_in x Q with Q <= 15, e.g. the output layer's [W_L|W_H|W_I] with
// <= 5 classes):  dX = dZ W^T  and  dW = X^T dZ  in ONE pass over X (gfx950).
// As two GEMMs (acm_gemm NT + TN with split-K) the N x f_in matrix X is streamed twice and dX once, in three
// launches (55 us at N = 168k, f_in = 64); both products are pure streaming (a few FMAs per loaded float), so here a
// 16-lane group takes one row -- lane m owns the four consecutive columns 4m..4m+3 of a 64-column chunk (one
// float4 load of X, one float4 store of dX, 256 contiguous bytes per row) -- with its 4 x Q slice of W and its
// 4 x Q accumulators of dW in registers.  dW partials: four row-groups of a wave by v_permlane swaps, the four
// waves through LDS, one slab per block, then a fixed-order column reduce (deterministic).
#include "acm_common.h"

namespace {

constexpr int PROJ_MAX_BLOCKS = 1024;

template <int Q>
__global__ __launch_bounds__(256) void proj_bwd_kernel(int n_rows, int f_in, const float* __restrict__ X, long ldx,
                                                       const float* __restrict__ dZ, long lddz,
                                                       const float* __restrict__ W0, const float* __restrict__ W1,
                                                       const float* __restrict__ W2, long ldw, float* __restrict__ dX,
                                                       long lddx, float* __restrict__ partial) {
    constexpr int F = Q / 3;                    // W = [W0 | W1 | W2], each f_in x F (the layer's three weights, unpacked)
    __shared__ float red[4][64 * Q];            // per wave: the wave's dW slice of the current chunk
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, m = lane & 15;
    const bool vec = ((((uintptr_t)X | (uintptr_t)dX) & 15) == 0) && (ldx % 4 == 0) && (lddx % 4 == 0);
    for (int j0 = 0; j0 < f_in; j0 += 64) {
        const int jb = j0 + 4 * m;              // first of this lane's four columns
        // this chunk of [W0 | W1 | W2] through LDS (coalesced, once per block; `red` is free until the end of the chunk),
        // then the lane's 4 x Q slice into registers
        for (int e = threadIdx.x; e < 64 * Q; e += 256) {
            const int j = j0 + e / Q, q = e % Q;
            const float* Wc = q < F ? W0 : (q < 2 * F ? W1 : W2);
            red[0][e] = j < f_in ? Wc[(long)j * ldw + (q % F)] : 0.f;
        }
        __syncthreads();
        float w[4][Q], acc[4][Q];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                w[i][q] = red[0][(4 * m + i) * Q + q];
                acc[i][q] = 0.f;
            }
        __syncthreads();                        // everyone has its slice before `red` is reused for the reduction
        const bool full = vec && jb + 3 < f_in;
        for (int row = (blockIdx.x * 4 + wave) * 4 + g; row < n_rows; row += gridDim.x * 16) {
            float dz[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) dz[q] = dZ[(long)row * lddz + q];
            float x[4];
            if (full) {
                const float4 v = *reinterpret_cast<const float4*>(X + (long)row * ldx + jb);
                x[0] = v.x, x[1] = v.y, x[2] = v.z, x[3] = v.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = (jb + i < f_in) ? X[(long)row * ldx + jb + i] : 0.f;
            }
            float dx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dx[i] = fmaf(dz[q], w[i][q], dx[i]);
                    acc[i][q] = fmaf(x[i], dz[q], acc[i][q]);
                }
            if (full) {
                *reinterpret_cast<float4*>(dX + (long)row * lddx + jb) = make_float4(dx[0], dx[1], dx[2], dx[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (jb + i < f_in) dX[(long)row * lddx + jb + i] = dx[i];
            }
        }
        // wave: sum the four row-groups; block: the four waves in a fixed order
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const float s = acm_cross_row_sum(acc[i][q]);
                if (g == 0) red[wave][(4 * m + i) * Q + q] = s;
            }
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * Q; e += 256) {
            const int j = j0 + e / Q;
            // slab in groups of 32 elements, partial[q / 32][block][q % 32] (acm_reduce_seg_t.elem_stride): see agg_bwd_kernel
            if (j < f_in) {
                const long q = (long)j * Q + (e % Q);
                partial[((q >> 5) * gridDim.x + blockIdx.x) * 32 + (q & 31)] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
            }
        }
        __syncthreads();
    }
}

// Forward of the same skinny projection: Z = relu?(X [W0 | W1 | W2]) for F <= 8 output columns per weight, straight from
// the three weight matrices (no packed copy), written as the two matrices the narrow fused layer wants: columns
// [0, 2F) -> Zlh (the gathered block [Z_L | Z_H]), [2F, 3F) -> Zi.  A 16-lane group per row, lane = 4 consecutive input
// columns (float4), the lane's 4 x 3F slice of the weights in registers, 3F group reductions per row (DPP).  One pass
// over X at stream speed; the MFMA GEMM spends 18 us on the 168 114 x 64 x 6 case, this one 11.
template <int F>
__global__ __launch_bounds__(256) void proj_fwd_kernel(int n_rows, int f_in, const float* __restrict__ X, long ldx,
                                                       const float* __restrict__ W0, const float* __restrict__ W1,
                                                       const float* __restrict__ W2, long ldw, int relu,
                                                       float* __restrict__ Zlh, long ld_lh, float* __restrict__ Zi, long ld_i,
                                                       int h_col) {
    constexpr int Q = 3 * F;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, m = lane & 15;
    const bool vec = (((uintptr_t)X & 15) == 0) && (ldx % 4 == 0);
    const bool out_vec = (((uintptr_t)Zlh & 15) == 0) && (ld_lh % 4 == 0) && (((uintptr_t)Zi & 7) == 0) && (ld_i % 2 == 0);
    const int chunks = (f_in + 63) / 64;
    // the first 64 input columns of [W0 | W1 | W2]: staged once per block through LDS (coalesced), then the lane's
    // 4 x Q slice stays in registers (per-lane global loads of it cost as much as the whole stream of X)
    __shared__ float wl[64 * Q];
    for (int e = threadIdx.x; e < 64 * Q; e += 256) {
        const int j = e / Q, q = e % Q;
        const float* Wc = q < F ? W0 : (q < 2 * F ? W1 : W2);
        wl[e] = j < f_in ? Wc[(long)j * ldw + (q % F)] : 0.f;
    }
    __syncthreads();
    float w0[4][Q];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < Q; ++q) w0[i][q] = wl[(4 * m + i) * Q + q];
    for (int row = (blockIdx.x * 4 + wave) * 4 + g; row < n_rows; row += gridDim.x * 16) {
        float part[Q];
        {
            float x[4];
            const int jb = 4 * m;
            if (vec && jb + 3 < f_in) {
                const float4 v = *reinterpret_cast<const float4*>(X + (long)row * ldx + jb);
                x[0] = v.x, x[1] = v.y, x[2] = v.z, x[3] = v.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = (jb + i < f_in) ? X[(long)row * ldx + jb + i] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < Q; ++q)
                part[q] = fmaf(x[3], w0[3][q], fmaf(x[2], w0[2][q], fmaf(x[1], w0[1][q], x[0] * w0[0][q])));
        }
        for (int ch = 1; ch < chunks; ++ch) {    // f_in > 64: further chunks read their weights through the cache
            const int jb = ch * 64 + 4 * m;
            float x[4];
            if (vec && jb + 3 < f_in) {
                const float4 v = *reinterpret_cast<const float4*>(X + (long)row * ldx + jb);
                x[0] = v.x, x[1] = v.y, x[2] = v.z, x[3] = v.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = (jb + i < f_in) ? X[(long)row * ldx + jb + i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = jb + i < f_in;
                const long wrow = (long)(ok ? jb + i : 0) * ldw;
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const float* Wc = q < F ? W0 : (q < 2 * F ? W1 : W2);
                    const float wv = ok ? Wc[wrow + (q % F)] : 0.f;          // 3 f_in F floats in total: L1 / K-cache resident
                    part[q] = fmaf(x[i], wv, part[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            part[q] = acm_group_sum<16>(part[q]);               // every lane of the group ends with the total
            if (relu) part[q] = fmaxf(part[q], 0.f);
        }
        if (F == 2 && out_vec && h_col == F) {                  // 16 + 8 bytes per row: two stores by the group leader
            if (m == 0) {
                *reinterpret_cast<float4*>(Zlh + (long)row * ld_lh) = make_float4(part[0], part[1], part[2], part[3]);
                *reinterpret_cast<float2*>(Zi + (long)row * ld_i) = make_float2(part[4], part[5]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (m == (q & 15)) {
                    if (q < F) Zlh[(long)row * ld_lh + q] = part[q];
                    else if (q < 2 * F) Zlh[(long)row * ld_lh + h_col + (q - F)] = part[q];       // Z_H starts at column h_col
                    else Zi[(long)row * ld_i + (q - 2 * F)] = part[q];
                }
        }
    }
}

int proj_blocks(int64_t n_rows) {
    int64_t nb = (n_rows + 15) / 16;
    if (nb > PROJ_MAX_BLOCKS) nb = PROJ_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    return (int)nb;
}

}  // namespace

extern "C" int acm_proj_fwd(int64_t n_rows, int64_t f_in, int f_out, const float* X, int64_t ldx, const float* w_low,
                            const float* w_high, const float* w_mlp, int64_t ldw, int relu, float* Z_lh, int64_t ld_lh,
                            float* Z_i, int64_t ld_i, acm_stream_t stream) {
    return acm_proj_fwd_at(n_rows, f_in, f_out, X, ldx, w_low, w_high, w_mlp, ldw, relu, Z_lh, ld_lh, f_out, Z_i, ld_i, stream);
}

extern "C" int acm_proj_fwd_at(int64_t n_rows, int64_t f_in, int f_out, const float* X, int64_t ldx, const float* w_low,
                               const float* w_high, const float* w_mlp, int64_t ldw, int relu, float* Z_lh, int64_t ld_lh,
                               int64_t h_col, float* Z_i, int64_t ld_i, acm_stream_t stream) {
    ACM_REQUIRE(X && w_low && w_high && w_mlp && Z_lh && Z_i, ACM_EINVAL, "acm_proj_fwd: NULL pointer");
    ACM_REQUIRE(f_out >= 1 && f_out <= 8, ACM_EUNSUPPORTED, "acm_proj_fwd: f_out = %d (1..8; use acm_gemm otherwise)", f_out);
    ACM_REQUIRE(n_rows >= 0 && n_rows < INT32_MAX && f_in >= 0 && f_in < INT32_MAX && ldx >= f_in && ldw >= f_out &&
                    h_col >= f_out && ld_lh >= h_col + f_out && ld_i >= f_out, ACM_ESHAPE,
                "acm_proj_fwd: bad sizes / leading dimensions");
    if (n_rows == 0) return ACM_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = proj_blocks(n_rows);
#define ACM_PF(Fv)                                                                                                  \
    hipLaunchKernelGGL((proj_fwd_kernel<Fv>), dim3(nblk), dim3(256), 0, s, (int)n_rows, (int)f_in, X, (long)ldx, w_low, \
                       w_high, w_mlp, (long)ldw, relu, Z_lh, (long)ld_lh, Z_i, (long)ld_i, (int)h_col)
    switch (f_out) {
        case 1: ACM_PF(1); break;
        case 2: ACM_PF(2); break;
        case 3: ACM_PF(3); break;
        case 4: ACM_PF(4); break;
        case 5: ACM_PF(5); break;
        case 6: ACM_PF(6); break;
        case 7: ACM_PF(7); break;
        default: ACM_PF(8); break;
    }
#undef ACM_PF
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}

extern "C" int acm_proj_bwd_workspace_bytes(int64_t n_rows, int64_t f_in, int n_out, size_t* bytes) {
    ACM_REQUIRE(bytes, ACM_EINVAL, "acm_proj_bwd_workspace_bytes: NULL argument");
    ACM_REQUIRE(n_rows >= 0 && f_in >= 0 && n_out >= 1, ACM_ESHAPE, "acm_proj_bwd_workspace_bytes: bad sizes");
    ACM_REQUIRE(n_out <= 15 && n_out % 3 == 0, ACM_EUNSUPPORTED,
                "acm_proj_bwd: n_out = %d (supported: 3, 6, 9, 12, 15; use acm_gemm otherwise)", n_out);
    *bytes = (size_t)proj_blocks(n_rows) * (((size_t)f_in * (size_t)n_out + 31) / 32 * 32) * sizeof(float);     // whole groups of 32
    return ACM_OK;
}

extern "C" int acm_proj_bwd(int64_t n_rows, int64_t f_in, int n_out, const float* X, int64_t ldx, const float* dZ,
                            int64_t lddz, const float* w_low, const float* w_high, const float* w_mlp, int64_t ldw,
                            float* dX, int64_t lddx, float* dW,
                            int64_t lddw, int64_t dw_col_block, int64_t dw_block_stride, void* workspace,
                            size_t workspace_bytes, acm_reduce_list_t* defer, acm_stream_t stream) {
    size_t need = 0;
    int st = acm_proj_bwd_workspace_bytes(n_rows, f_in, n_out, &need);
    if (st != ACM_OK) return st;
    ACM_REQUIRE(X && dZ && w_low && w_high && w_mlp && dX && dW, ACM_EINVAL, "acm_proj_bwd: NULL pointer");
    ACM_REQUIRE(n_rows < INT32_MAX && f_in < INT32_MAX && ldx >= f_in && lddx >= f_in && lddz >= n_out && ldw >= n_out / 3 &&
                    dw_col_block >= 0 && lddw >= (dw_col_block ? (dw_col_block < n_out ? dw_col_block : n_out) : n_out),
                ACM_ESHAPE, "acm_proj_bwd: bad sizes / leading dimensions");
    if (f_in == 0) return ACM_OK;
    ACM_REQUIRE(workspace && workspace_bytes >= need, ACM_ENOMEM, "acm_proj_bwd: workspace %zu B < required %zu B",
                workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    const int nblk = proj_blocks(n_rows);
#define ACM_PROJ(Qv)                                                                                              \
    hipLaunchKernelGGL((proj_bwd_kernel<Qv>), dim3(nblk), dim3(256), 0, s, (int)n_rows, (int)f_in, X, (long)ldx, dZ, \
                       (long)lddz, w_low, w_high, w_mlp, (long)ldw, dX, (long)lddx, partial)
    switch (n_out) {
        case 3: ACM_PROJ(3); break;
        case 6: ACM_PROJ(6); break;
        case 9: ACM_PROJ(9); break;
        case 12: ACM_PROJ(12); break;
        default: ACM_PROJ(15); break;
    }
#undef ACM_PROJ
    ACM_CHECK_HIP(hipGetLastError());
    // dW[j][q] = sum_b partial[(j * n_out + q) / 32][b][(j * n_out + q) % 32], optionally in the column-block layout of dW
    const acm_reduce_seg_t seg = {partial, nblk, 32, 0, (int32_t)(f_in * n_out), dW, n_out,
                                  (int32_t)dw_col_block, lddw, dw_block_stride, nblk * 32, 0};
    return acm_reduce_emit(defer, &seg, 1, s);
}
