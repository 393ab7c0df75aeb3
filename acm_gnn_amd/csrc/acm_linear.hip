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
