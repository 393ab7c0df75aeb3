// Fused step tail: row-wise log-softmax + weighted NLL + gradient, one pass (gfx950).
// One thread per row (n_classes <= 64 values, consecutive rows => coalesced), per-block LDS tree for
// the loss partial, second launch adds the block partials in index order: deterministic.
#include <math.h>

#include "acm_common.h"

namespace {

__global__ __launch_bounds__(256) void nll_rows_kernel(int n, int C, const float* __restrict__ z, long ldz,
                                                       const int64_t* __restrict__ y, const float* __restrict__ w,
                                                       float* __restrict__ dz, long ldd, float* __restrict__ partial) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        acc += acm_nll_row(C, z + (long)i * ldz, (int)y[i], w[i], dz + (long)i * ldd);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k >= 1; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

int nll_blocks(int64_t n) {
    int64_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    return (int)nb;
}

}  // namespace

extern "C" int acm_nll_loss_workspace_bytes(int64_t n_rows, size_t* bytes) {
    ACM_REQUIRE(bytes, ACM_EINVAL, "acm_nll_loss_workspace_bytes: NULL argument");
    ACM_REQUIRE(n_rows >= 0, ACM_ESHAPE, "acm_nll_loss_workspace_bytes: negative size");
    *bytes = (size_t)nll_blocks(n_rows) * sizeof(float);
    return ACM_OK;
}

extern "C" int acm_nll_loss(int64_t n_rows, int n_classes, const float* logits, int64_t ld_logits,
                            const int64_t* labels, const float* row_weight, float* loss, float* dlogits,
                            int64_t ld_dlogits, void* workspace, size_t workspace_bytes, acm_reduce_list_t* defer,
                            acm_stream_t stream) {
    ACM_REQUIRE(logits && labels && row_weight && loss && dlogits, ACM_EINVAL, "acm_nll_loss: NULL pointer");
    ACM_REQUIRE(n_rows >= 0 && n_rows < INT32_MAX && n_classes >= 1, ACM_ESHAPE, "acm_nll_loss: bad sizes");
    ACM_REQUIRE(n_classes <= 64, ACM_EUNSUPPORTED, "acm_nll_loss: %d classes > 64", n_classes);
    ACM_REQUIRE(ld_logits >= n_classes && ld_dlogits >= n_classes, ACM_ESHAPE, "acm_nll_loss: leading dimension too small");
    const int nblk = nll_blocks(n_rows);
    ACM_REQUIRE(workspace && workspace_bytes >= nblk * sizeof(float), ACM_ENOMEM, "acm_nll_loss: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(nll_rows_kernel, dim3(nblk), dim3(256), 0, st, (int)n_rows, n_classes, logits, (long)ld_logits,
                       labels, row_weight, dlogits, (long)ld_dlogits, partial);
    ACM_CHECK_HIP(hipGetLastError());
    const acm_reduce_seg_t seg = {partial, nblk, 1, 0, 1, loss, 1, 0, 0, 0};
    return acm_reduce_emit(defer, &seg, 1, st);
}
