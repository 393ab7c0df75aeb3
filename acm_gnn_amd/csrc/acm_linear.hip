// Row-local pieces of the ACM-GCN++ residual branch  xX = dropout(relu(Linear(x)))
// (ACM-Geometric/models.py:26-27,55-56; ACM-Pytorch/models/models.py:50-53,150-152):
//   acm_bias_act      Y <- dropout(relu?(Y + b)) in place -- the epilogue of the CSR-feature route, where the product
//                     X_csr W^T comes out of acm_spmm_v (dense features get it inside the GEMM: acm_linear_fwd)
//   acm_bias_act_bwd  G = dL/d(pre-activation) = dY * keep / (1 - p) * [pre > 0]  and  db = column sums of G.
// The ReLU mask and the dropout mask are both read off the forward's OUTPUT: Y > 0 iff pre > 0 and kept, and where
// Y == 0 the gradient is 0 either way -- no mask tensor, no Philox replay.
#include "acm_common.h"

namespace {

__global__ __launch_bounds__(256) void bias_act_kernel(long n_rows, int f, float* __restrict__ y, long ldy,
                                                       const float* __restrict__ bias, int relu, acm_dropout_t drop) {
    const AcmDropCtx dc = acm_drop_ctx(drop);
    const long total = n_rows * f;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long r = q / f;
        const int c = (int)(q - r * f);
        float v = y[r * ldy + c];
        if (bias) v += bias[c];
        if (relu) v = fmaxf(v, 0.f);
        if (dc.on) v *= acm_drop1(dc, r, c);
        y[r * ldy + c] = v;
    }
}

// One thread per column (f <= 256), a block walks a stripe of rows: coalesced along the row, fixed summation order.
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(long n_rows, int f, const float* __restrict__ y, long ldy,
                                                           const float* __restrict__ dy, long lddy, float inv_keep,
                                                           int relu, float* __restrict__ g, long ldg,
                                                           float* __restrict__ partial) {
    const int c = threadIdx.x;
    float acc = 0.f;
    if (c < f) {
        for (long r = blockIdx.x; r < n_rows; r += gridDim.x) {
            const float out = y[r * ldy + c];
            float v = dy[r * lddy + c];
            // relu: kept and active iff Y > 0.  No relu but dropout: dropped iff Y == 0 (a kept exact zero is a
            // measure-zero event and its gradient is lost).  Neither: G = dY.
            if (relu) v = out > 0.f ? v * inv_keep : 0.f;
            else if (inv_keep != 1.f) v = out != 0.f ? v * inv_keep : 0.f;
            g[r * ldg + c] = v;
            acc += v;
        }
        partial[(long)blockIdx.x * f + c] = acc;
    }
}

int bias_bwd_blocks(int64_t n_rows) { return (int)(n_rows < 1024 ? (n_rows < 1 ? 1 : n_rows) : 1024); }

// ---- the residual Linear on a NARROW dense input (f_in <= 16: the raw features; 7 columns on twitch-gamer), as streaming
// kernels.  A wave takes chunks of EIGHT consecutive rows (chunk = wave index + k * waves): the 8 x FI block of X goes through
// 256-512 B of wave-private LDS (one coalesced load, then broadcast reads: no barrier, one wave, in-order LDS), lane = output
// column, 8 (forward: stores) or 16 (backward: Y and dY) row accesses in flight.
// (load and LDS store are separate so that a wave requests the NEXT chunk's block -- and its Y / dY / add rows -- before it
//  works on the one in hand: a wave has only two or three chunks, every exposed round trip counts)
template <int FI>
__device__ __forceinline__ void load_x_rows(float (&xe)[FI / 8], const float* __restrict__ x, long ldx, long r0, long n_rows, int f_in,
                                            int lane) {
#pragma unroll
    for (int k = 0; k < FI / 8; ++k) {
        const int e = 64 * k + lane, u = e / FI, f = e % FI;
        const long r = r0 + u;
        const bool in = r < n_rows && f < f_in;
        const float v = x[(in ? r : 0) * ldx + (in ? f : 0)];
        xe[k] = in ? v : 0.f;
    }
}
template <int FI>
__device__ __forceinline__ void store_x_rows(float* xs, const float (&xe)[FI / 8], int lane) {
#pragma unroll
    for (int k = 0; k < FI / 8; ++k) xs[64 * k + lane] = xe[k];
}

// Forward: Y = dropout(relu(X W^T + b)).  The product is an fmaf chain in k order from zero, bias -> ReLU -> dropout after it: the
// arithmetic (and the bits) of the GEMM route with its epilogue.  The counter-based mask: one Philox call yields the factors
// of columns i, i + 16, i + 32, i + 48 of a row (acm_drop4), so lane row g of the wave draws for row g of a half chunk and
// the 64 factors of a row meet in LDS -- a quarter of the per-element calls of the GEMM epilogue.
template <int FI>
__global__ __launch_bounds__(256) void linear_fwd_narrow_kernel(long n_rows, int f_in, int f_out, const float* __restrict__ x, long ldx,
                                                                const float* __restrict__ w, long ldw, const float* __restrict__ bias,
                                                                int relu, acm_dropout_t drop, const float* add, long ld_add,
                                                                float* y, long ldy) {
    __shared__ __attribute__((aligned(16))) float xs_all[4][8 * FI];
    __shared__ float fac_all[4][8 * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    float* xs = xs_all[wv];
    float* fac = fac_all[wv];
    const long n_waves = (long)gridDim.x * 4, gw = (long)blockIdx.x * 4 + wv;
    const AcmDropCtx dc = acm_drop_ctx(drop);
    const long n_chunks = (n_rows + 7) / 8;
    for (int c0 = 0; c0 < f_out; c0 += 64) {
        const int o = c0 + lane;
        const bool ok = o < f_out;
        float wr[FI];
#pragma unroll
        for (int f = 0; f < FI; ++f) wr[f] = (ok && f < f_in) ? w[(long)o * ldw + f] : 0.f;
        const float b = (ok && bias) ? bias[o] : 0.f;
        float xe[FI / 8], av[8];
        auto fetch = [&](long ch) {                      // chunk ch's block of X and rows of `add` (clamped addresses, no branches)
            const long r0 = ch * 8;
            load_x_rows<FI>(xe, x, ldx, r0, n_rows, f_in, lane);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long r = (r0 + u < n_rows) ? r0 + u : 0;
                av[u] = add ? add[r * ld_add + (ok ? o : 0)] : 0.f;
            }
        };
        if (gw < n_chunks) fetch(gw);
        for (long ch = gw; ch < n_chunks; ch += n_waves) {
            const long r0 = ch * 8;
            store_x_rows<FI>(xs, xe, lane);
            float ac[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ac[u] = av[u];
            if (ch + n_waves < n_chunks) fetch(ch + n_waves);
            if (dc.on) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float f4[4];
                    acm_drop4(dc, r0 + 4 * h + g, i + 16 * (c0 >> 6), f4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) fac[(4 * h + g) * 64 + 16 * q + i] = f4[q];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long r = r0 + u;
                float v = 0.f;
#pragma unroll
                for (int f = 0; f < FI; ++f)
                    v = fmaf(xs[u * FI + f], wr[f], v);       // (columns beyond f_in: zeros in xs and wr)
                v += b;
                if (relu) v = fmaxf(v, 0.f);
                if (dc.on) v *= fac[u * 64 + lane];
                if (ok && r < n_rows) y[r * ldy + o] = add ? v + ac[u] : v;   // (y may be add itself: a chunk's rows are read before they are
                                                                     //  written, and only this wave touches them)
            }
        }
    }
}

// Backward: dW = G^T X and db = column sums of G in ONE pass over (Y, dY, X), G = bias_act_bwd_kernel's gradient formed in
// registers and never stored -- the two launches it replaces wrote and re-read the [n, f_out] matrix G (2 x 43 MB on the
// twitch graph) and ran 64 of 256 threads.  Fixed summation order: deterministic.
// partial: [groups of 32 columns][block][32], column = f * f_out + o (f = f_in: the bias).
// RECOMPUTE: the masks are not read off Y (which the caller no longer has: acm_linear_fwd_add added another tensor to it) but
// formed again -- the pre-activation by the forward's own fmaf chain (same bits, so the same sign), the dropout factors from
// the counter (same Philox call per four columns): the pass reads dY and X only.
struct LinRecompute {
    const float* w;
    long ldw;
    const float* bias;
    acm_dropout_t drop;
};
template <int FI, bool RECOMPUTE>
__global__ __launch_bounds__(256) void linear_bwd_narrow_kernel(long n_rows, int f_in, int f_out, const float* __restrict__ x, long ldx,
                                                                const float* __restrict__ y, long ldy, const float* __restrict__ dy,
                                                                long lddy, float inv_keep, int relu, LinRecompute rc,
                                                                float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float xs_all[4][8 * FI];
    __shared__ float red[4][64 * (FI + 1)];
    __shared__ float fac_all[RECOMPUTE ? 4 : 1][RECOMPUTE ? 8 * 64 : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* xs = xs_all[wv];
    float* fac = fac_all[RECOMPUTE ? wv : 0];
    const AcmDropCtx dc = acm_drop_ctx(rc.drop);
    const long n_waves = (long)gridDim.x * 4, gw = (long)blockIdx.x * 4 + wv;
    const long gstride = (long)gridDim.x * 32;
    const long n_chunks = (n_rows + 7) / 8;
    for (int c0 = 0; c0 < f_out; c0 += 64) {
        const int o = c0 + lane;
        const bool ok = o < f_out;
        const int oc = ok ? o : 0;
        float acc[FI], accb = 0.f, wr[FI];
#pragma unroll
        for (int f = 0; f < FI; ++f) {
            acc[f] = 0.f;
            wr[f] = (RECOMPUTE && ok && f < f_in) ? rc.w[(long)o * rc.ldw + f] : 0.f;
        }
        const float bo = (RECOMPUTE && ok && rc.bias) ? rc.bias[o] : 0.f;
        float xe[FI / 8], dn[8], yn[8];
        auto fetch = [&](long ch) {                      // chunk ch's block of X and its rows of dY (and Y): clamped addresses
            const long r0 = ch * 8;
            load_x_rows<FI>(xe, x, ldx, r0, n_rows, f_in, lane);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long r = (r0 + u < n_rows) ? r0 + u : 0;
                dn[u] = dy[r * lddy + oc];
                yn[u] = RECOMPUTE ? 0.f : y[r * ldy + oc];
            }
        };
        if (gw < n_chunks) fetch(gw);
        for (long ch = gw; ch < n_chunks; ch += n_waves) {
            const long r0 = ch * 8;
            float gv[8], dv[8], yv[8];
            store_x_rows<FI>(xs, xe, lane);
#pragma unroll
            for (int u = 0; u < 8; ++u) dv[u] = dn[u], yv[u] = yn[u];
            if (ch + n_waves < n_chunks) fetch(ch + n_waves);
            if (RECOMPUTE) {
                if (dc.on) {
                    const int g = lane >> 4, i = lane & 15;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float f4[4];
                        acm_drop4(dc, r0 + 4 * h + g, i + 16 * (c0 >> 6), f4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) fac[(4 * h + g) * 64 + 16 * q + i] = f4[q];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = r0 + u < n_rows;
                float v = dv[u];
                if (RECOMPUTE) {
                    float pre = 0.f;
#pragma unroll
                    for (int f = 0; f < FI; ++f)
                        pre = fmaf(xs[u * FI + f], wr[f], pre);   // (columns beyond f_in: zeros in xs and wr)
                    pre += bo;
                    if (relu) v = pre > 0.f ? v : 0.f;
                    if (dc.on) v *= fac[u * 64 + lane];
                } else {
                    const float out = yv[u];
                    // (bias_act_bwd_kernel's formula: both masks are read off the forward's output)
                    if (relu) v = out > 0.f ? v * inv_keep : 0.f;
                    else if (inv_keep != 1.f) v = out != 0.f ? v * inv_keep : 0.f;
                }
                gv[u] = (in && ok) ? v : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                accb += gv[u];
#pragma unroll
                for (int f = 0; f < FI; ++f)
                    acc[f] = fmaf(gv[u], xs[u * FI + f], acc[f]);   // (columns beyond f_in: zeros in xs; never written out)
            }
        }
#pragma unroll
        for (int f = 0; f < FI; ++f) red[wv][f * 64 + lane] = acc[f];
        red[wv][FI * 64 + lane] = accb;
        __syncthreads();
        for (int q = threadIdx.x; q < 64 * (FI + 1); q += 256) {
            const int f = q >> 6, l = q & 63;
            const int fo = f == FI ? f_in : f;           // the bias sits behind the f_in weight rows
            if ((f < f_in || f == FI) && c0 + l < f_out) {
                const float s = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
                const int col = fo * f_out + c0 + l;
                partial[(long)(col >> 5) * gstride + (long)blockIdx.x * 32 + (col & 31)] = s;
            }
        }
        __syncthreads();
    }
}

int linear_bwd_blocks(int64_t n_rows) {
    const int64_t b = (n_rows + 63) / 64;                // >= 16 rows per wave before another block pays
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int acm_bias_act(int64_t n_rows, int f, float* Y, int64_t ldy, const float* bias, int relu,
                            const acm_dropout_t* drop, acm_stream_t stream) {
    ACM_REQUIRE(Y, ACM_EINVAL, "acm_bias_act: NULL argument");
    ACM_REQUIRE(n_rows >= 0 && f > 0 && ldy >= f, ACM_ESHAPE, "acm_bias_act: n_rows %lld f %d ldy %lld", (long long)n_rows, f,
                (long long)ldy);
    acm_dropout_t d = {0.f, 0, 0, nullptr, 0};
    if (drop) d = *drop;
    ACM_REQUIRE(d.p == 0.f || (d.p > 0.f && d.p < 1.f && d.step), ACM_EINVAL, "acm_bias_act: bad dropout spec");
    if (n_rows == 0) return ACM_OK;
    long blocks = (n_rows * f + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bias_act_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long)n_rows, f, Y,
                       (long)ldy, bias, relu, d);
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}

extern "C" int acm_bias_act_bwd_workspace_bytes(int64_t n_rows, int f, size_t* bytes) {
    ACM_REQUIRE(bytes, ACM_EINVAL, "acm_bias_act_bwd_workspace_bytes: NULL argument");
    ACM_REQUIRE(n_rows >= 0 && f > 0 && f <= 256, ACM_EUNSUPPORTED, "acm_bias_act_bwd: f %d outside 1..256", f);
    *bytes = (size_t)bias_bwd_blocks(n_rows) * (size_t)f * sizeof(float);
    return ACM_OK;
}

extern "C" int acm_bias_act_bwd(int64_t n_rows, int f, const float* Y, int64_t ldy, const float* dY, int64_t lddy,
                                float keep_scale, int relu, float* G, int64_t ldg, float* d_bias, void* workspace,
                                size_t workspace_bytes, acm_reduce_list_t* defer, acm_stream_t stream) {
    ACM_REQUIRE(Y && dY && G && d_bias, ACM_EINVAL, "acm_bias_act_bwd: NULL argument");
    size_t need = 0;
    int st = acm_bias_act_bwd_workspace_bytes(n_rows, f, &need);
    if (st != ACM_OK) return st;
    ACM_REQUIRE(ldy >= f && lddy >= f && ldg >= f && keep_scale >= 1.f, ACM_ESHAPE, "acm_bias_act_bwd: leading dimensions / scale");
    ACM_REQUIRE(workspace && workspace_bytes >= need, ACM_ENOMEM, "acm_bias_act_bwd: workspace %zu B < required %zu B",
                workspace_bytes, need);
    const int nblk = bias_bwd_blocks(n_rows);
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(bias_act_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (long)n_rows, f, Y, (long)ldy, dY,
                       (long)lddy, keep_scale, relu, G, (long)ldg, partial);
    ACM_CHECK_HIP(hipGetLastError());
    const acm_reduce_seg_t seg = {partial, nblk, f, 0, f, d_bias, f, 0, 0, 0};
    return acm_reduce_emit(defer, &seg, 1, (hipStream_t)stream);
}

// dW[o][f] <- column f * f_out + o: inner = f_out columns per "row" f, destination o * lddw + f (col_block 1, block_stride lddw)
static int linear_bwd_segments(float* partial, int nblk, int f_in, int f_out, float* dW, int64_t lddw, float* d_bias,
                               acm_reduce_list_t* defer, hipStream_t s) {
    const acm_reduce_seg_t segs[2] = {{partial, nblk, 32, 0, f_in * f_out, dW, f_out, 1, 1, lddw, nblk * 32, 0},
                                      {partial, nblk, 32, f_in * f_out, f_out, d_bias, f_out, 0, 0, 0, nblk * 32, 0}};
    return acm_reduce_emit(defer, segs, 2, s);
}

extern "C" int acm_linear_bwd_workspace_bytes(int64_t n_rows, int f_in, int f_out, size_t* bytes) {
    ACM_REQUIRE(bytes, ACM_EINVAL, "acm_linear_bwd_workspace_bytes: NULL argument");
    ACM_REQUIRE(n_rows >= 0 && f_in >= 1 && f_in <= 16 && f_out >= 1 && f_out <= 256, ACM_EUNSUPPORTED,
                "acm_linear_bwd: f_in %d (1..16), f_out %d (1..256)", f_in, f_out);
    const size_t groups = ((size_t)(f_in + 1) * f_out + 31) / 32;
    *bytes = groups * (size_t)linear_bwd_blocks(n_rows) * 32 * sizeof(float);
    return ACM_OK;
}

extern "C" int acm_linear_bwd(int64_t n_rows, int f_in, int f_out, const float* X, int64_t ldx, const float* Y, int64_t ldy,
                              const float* dY, int64_t lddy, float keep_scale, int relu, float* dW, int64_t lddw, float* d_bias,
                              void* workspace, size_t workspace_bytes, acm_reduce_list_t* defer, acm_stream_t stream) {
    ACM_REQUIRE(X && Y && dY && dW && d_bias, ACM_EINVAL, "acm_linear_bwd: NULL argument");
    size_t need = 0;
    const int st = acm_linear_bwd_workspace_bytes(n_rows, f_in, f_out, &need);
    if (st != ACM_OK) return st;
    ACM_REQUIRE(ldx >= f_in && ldy >= f_out && lddy >= f_out && lddw >= f_in && keep_scale >= 1.f, ACM_ESHAPE,
                "acm_linear_bwd: leading dimensions / scale");
    ACM_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace) % 16 == 0, ACM_ENOMEM,
                "acm_linear_bwd: workspace %zu B < required %zu B (or not 16-byte aligned)", workspace_bytes, need);
    ACM_REQUIRE(n_rows < ((int64_t)1 << 40), ACM_EUNSUPPORTED, "acm_linear_bwd: %lld rows", (long long)n_rows);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = linear_bwd_blocks(n_rows);
    float* partial = (float*)workspace;
    if (n_rows == 0) ACM_CHECK_HIP(hipMemsetAsync(partial, 0, need, s));
    else {
        const LinRecompute none = {nullptr, 0, nullptr, {0.f, 0, 0, nullptr, 0}};
        if (f_in <= 8)
            hipLaunchKernelGGL((linear_bwd_narrow_kernel<8, false>), dim3(nblk), dim3(256), 0, s, (long)n_rows, f_in, f_out, X, (long)ldx, Y,
                               (long)ldy, dY, (long)lddy, keep_scale, relu, none, partial);
        else
            hipLaunchKernelGGL((linear_bwd_narrow_kernel<16, false>), dim3(nblk), dim3(256), 0, s, (long)n_rows, f_in, f_out, X, (long)ldx, Y,
                               (long)ldy, dY, (long)lddy, keep_scale, relu, none, partial);
    }
    ACM_CHECK_HIP(hipGetLastError());
    return linear_bwd_segments(partial, nblk, f_in, f_out, dW, lddw, d_bias, defer, s);
}

extern "C" int acm_linear_bwd_recompute(int64_t n_rows, int f_in, int f_out, const float* X, int64_t ldx, const float* W, int64_t ldw,
                                        const float* bias, int relu, const acm_dropout_t* drop, const float* dY, int64_t lddy,
                                        float* dW, int64_t lddw, float* d_bias, void* workspace, size_t workspace_bytes,
                                        acm_reduce_list_t* defer, acm_stream_t stream) {
    ACM_REQUIRE(X && W && dY && dW && d_bias, ACM_EINVAL, "acm_linear_bwd_recompute: NULL argument");
    size_t need = 0;
    const int st = acm_linear_bwd_workspace_bytes(n_rows, f_in, f_out, &need);
    if (st != ACM_OK) return st;
    acm_dropout_t d = {0.f, 0, 0, nullptr, 0};
    if (drop) d = *drop;
    ACM_REQUIRE(d.p == 0.f || (d.p > 0.f && d.p < 1.f && d.step), ACM_EINVAL, "acm_linear_bwd_recompute: bad dropout spec");
    ACM_REQUIRE(ldx >= f_in && ldw >= f_in && lddy >= f_out && lddw >= f_in, ACM_ESHAPE, "acm_linear_bwd_recompute: leading dimension too small");
    ACM_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace) % 16 == 0, ACM_ENOMEM,
                "acm_linear_bwd_recompute: workspace %zu B < required %zu B (or not 16-byte aligned)", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = linear_bwd_blocks(n_rows);
    float* partial = (float*)workspace;
    const LinRecompute rc = {W, (long)ldw, bias, d};
    if (n_rows == 0) ACM_CHECK_HIP(hipMemsetAsync(partial, 0, need, s));
    else if (f_in <= 8)
        hipLaunchKernelGGL((linear_bwd_narrow_kernel<8, true>), dim3(nblk), dim3(256), 0, s, (long)n_rows, f_in, f_out, X, (long)ldx, nullptr,
                           0L, dY, (long)lddy, 1.f, relu, rc, partial);
    else
        hipLaunchKernelGGL((linear_bwd_narrow_kernel<16, true>), dim3(nblk), dim3(256), 0, s, (long)n_rows, f_in, f_out, X, (long)ldx, nullptr,
                           0L, dY, (long)lddy, 1.f, relu, rc, partial);
    ACM_CHECK_HIP(hipGetLastError());
    return linear_bwd_segments(partial, nblk, f_in, f_out, dW, lddw, d_bias, defer, s);
}

// acm_linear_fwd's route for narrow inputs (called from acm_gemm.hip); ACM_EUNSUPPORTED: the GEMM route
int acm_linear_fwd_narrow(int64_t n_rows, int64_t f_in, int64_t f_out, const float* X, int64_t ldx, const float* W, int64_t ldw,
                          const float* bias, int relu, const acm_dropout_t* drop, float* Y, int64_t ldy, hipStream_t s,
                          const float* add, int64_t ld_add) {
    if (f_in < 1 || f_in > 16 || f_out < 1 || f_out > 256 || (n_rows < 1024 && !add)) return ACM_EUNSUPPORTED;
    ACM_REQUIRE(!add || ld_add >= f_out, ACM_ESHAPE, "acm_linear_fwd_add: ld_add too small");
    if (n_rows == 0) return ACM_OK;
    acm_dropout_t d = {0.f, 0, 0, nullptr, 0};
    if (drop) d = *drop;
    ACM_REQUIRE(d.p == 0.f || (d.p > 0.f && d.p < 1.f && d.step), ACM_EINVAL, "acm_linear_fwd: bad dropout spec");
    ACM_REQUIRE(ldx >= f_in && ldw >= f_in && ldy >= f_out, ACM_ESHAPE, "acm_linear_fwd: leading dimension too small");
    const int nblk = linear_bwd_blocks(n_rows);
    if (f_in <= 8)
        hipLaunchKernelGGL(linear_fwd_narrow_kernel<8>, dim3(nblk), dim3(256), 0, s, (long)n_rows, (int)f_in, (int)f_out, X, (long)ldx, W,
                           (long)ldw, bias, relu, d, add, (long)ld_add, Y, (long)ldy);
    else
        hipLaunchKernelGGL(linear_fwd_narrow_kernel<16>, dim3(nblk), dim3(256), 0, s, (long)n_rows, (int)f_in, (int)f_out, X, (long)ldx, W,
                           (long)ldw, bias, relu, d, add, (long)ld_add, Y, (long)ldy);
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}

// Y = add + dropout(relu(X W^T + b)): the ACM-GCN++ hidden activations fea1 + xX (ACM-Geometric/models.py:73) in the Linear's own
// launch -- no [n, f_out] tensor xX, no separate add.  Narrow dense inputs only (f_in <= 16, f_out <= 256).
extern "C" int acm_linear_fwd_add(int64_t n_rows, int f_in, int f_out, const float* X, int64_t ldx, const float* W, int64_t ldw,
                                  const float* bias, int relu, const acm_dropout_t* drop, const float* add, int64_t ld_add, float* Y,
                                  int64_t ldy, acm_stream_t stream) {
    ACM_REQUIRE(X && W && Y && add, ACM_EINVAL, "acm_linear_fwd_add: NULL argument");
    ACM_REQUIRE(n_rows >= 0, ACM_ESHAPE, "acm_linear_fwd_add: negative row count");
    const int st = acm_linear_fwd_narrow(n_rows, f_in, f_out, X, ldx, W, ldw, bias, relu, drop, Y, ldy, (hipStream_t)stream, add, ld_add);
    if (st == ACM_EUNSUPPORTED) acm_set_error("acm_linear_fwd_add: f_in %d (1..16), f_out %d (1..256)", f_in, f_out);
    return st;
}
