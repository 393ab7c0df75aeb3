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

// Evaluation metrics (acm_eval_metrics): one thread per row, S + 1 running sums per thread (accuracy per index set, NLL on one
// set), block tree in LDS, per-block partials, and the block that arrives LAST adds the partials in block order and resets
// the counter -- one launch, the same bits on every run.
constexpr int EVAL_MAX_SETS = 8;

__global__ __launch_bounds__(256) void eval_metrics_kernel(int n, int C, const float* __restrict__ z, long ldz,
                                                           const int64_t* __restrict__ y, const float* __restrict__ w, long ldw,
                                                           int S, int loss_set, float* __restrict__ out,
                                                           float* __restrict__ partial, int* __restrict__ arrive) {
    __shared__ float red[EVAL_MAX_SETS + 1][4];
    __shared__ int last;
    float acc[EVAL_MAX_SETS + 1];
#pragma unroll
    for (int s = 0; s <= EVAL_MAX_SETS; ++s) acc[s] = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float ws[EVAL_MAX_SETS];
        bool any = false;
#pragma unroll
        for (int s = 0; s < EVAL_MAX_SETS; ++s) {
            ws[s] = s < S ? w[(long)s * ldw + i] : 0.f;
            any = any || ws[s] != 0.f;
        }
        if (!any) continue;                              // (also: an unlabeled row, y = -1, is never looked up)
        const float* zi = z + (long)i * ldz;
        const int yi = (int)y[i];
        float m = zi[0];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float v = zi[c];
            if (v > m) m = v, arg = c;                   // strict: the first maximum wins, like torch.argmax
        }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(zi[c] - m);
        const float nll = (m + logf(se)) - zi[yi];
        const float hit = arg == yi ? 1.f : 0.f;
#pragma unroll
        for (int s = 0; s < EVAL_MAX_SETS; ++s) acc[s] = fmaf(ws[s], hit, acc[s]);
        float wl = 0.f;
#pragma unroll
        for (int s = 0; s < EVAL_MAX_SETS; ++s) wl = s == loss_set ? ws[s] : wl;
        acc[EVAL_MAX_SETS] = fmaf(wl, nll, acc[EVAL_MAX_SETS]);
    }
    // block sums: every wave adds its 64 lanes in registers (DPP: a fixed order), the four wave sums meet in LDS
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int s = 0; s <= EVAL_MAX_SETS; ++s) {
        if (s < S || s == EVAL_MAX_SETS) {                      // (S is uniform: no divergence)
            const float t = acm_group_sum<64>(acc[s]);
            if (lane == 0) red[s][wv] = t;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x <= S) {                     // slot S of a block's record = the loss sum
        const float* r = red[(int)threadIdx.x < S ? threadIdx.x : EVAL_MAX_SETS];
        __hip_atomic_store(partial + (long)blockIdx.x * (S + 1) + threadIdx.x, (r[0] + r[1]) + (r[2] + r[3]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int old = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = old == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // thread b takes block b's record (at most 256 blocks: eval_blocks), then the same fixed tree -- a serial walk of 256
    // cache-bypassing loads by one thread cost 38 us of the 43 us launch on the twitch-shaped graph
#pragma unroll
    for (int s = 0; s <= EVAL_MAX_SETS; ++s) {
        if (s < S || s == EVAL_MAX_SETS) {
            const int col = s < EVAL_MAX_SETS ? s : S;              // (slot S of a record = the loss sum)
            const float v = (int)threadIdx.x < (int)gridDim.x
                                ? __hip_atomic_load(partial + (long)threadIdx.x * (S + 1) + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            const float t = acm_group_sum<64>(v);
            if (lane == 0) red[s][wv] = t;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x <= S) {
        const float* r = red[(int)threadIdx.x < S ? threadIdx.x : EVAL_MAX_SETS];
        out[threadIdx.x] = (r[0] + r[1]) + (r[2] + r[3]);
    }
    if (threadIdx.x == 0) __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int eval_blocks(int64_t n) {
    int64_t nb = (n + 255) / 256;
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    return (int)nb;
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

extern "C" int acm_eval_metrics_workspace_bytes(int64_t n_rows, int n_sets, size_t* bytes) {
    ACM_REQUIRE(bytes, ACM_EINVAL, "acm_eval_metrics_workspace_bytes: NULL argument");
    ACM_REQUIRE(n_rows >= 0 && n_sets >= 1 && n_sets <= EVAL_MAX_SETS, ACM_ESHAPE, "acm_eval_metrics_workspace_bytes: bad sizes");
    *bytes = ((size_t)eval_blocks(n_rows) * (size_t)(n_sets + 1) + 1) * sizeof(float);      // partials + the arrival counter
    return ACM_OK;
}

extern "C" int acm_eval_metrics(int64_t n_rows, int n_classes, const float* logits, int64_t ld_logits, const int64_t* labels,
                                const float* weights, int64_t ld_weights, int n_sets, int loss_set, float* out,
                                void* workspace, size_t workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(logits && labels && weights && out, ACM_EINVAL, "acm_eval_metrics: NULL pointer");
    ACM_REQUIRE(n_rows >= 0 && n_rows < INT32_MAX && n_classes >= 1, ACM_ESHAPE, "acm_eval_metrics: bad sizes");
    ACM_REQUIRE(n_classes <= 64, ACM_EUNSUPPORTED, "acm_eval_metrics: %d classes > 64", n_classes);
    ACM_REQUIRE(n_sets >= 1 && n_sets <= EVAL_MAX_SETS && loss_set >= 0 && loss_set < n_sets, ACM_ESHAPE,
                "acm_eval_metrics: %d index sets (1..%d), loss set %d", n_sets, EVAL_MAX_SETS, loss_set);
    ACM_REQUIRE(ld_logits >= n_classes && ld_weights >= n_rows, ACM_ESHAPE, "acm_eval_metrics: leading dimension too small");
    const int nblk = eval_blocks(n_rows);
    const size_t need = ((size_t)nblk * (size_t)(n_sets + 1) + 1) * sizeof(float);
    ACM_REQUIRE(workspace && workspace_bytes >= need, ACM_ENOMEM, "acm_eval_metrics: workspace %zu B < required %zu B", workspace_bytes, need);
    float* partial = (float*)workspace;
    int* arrive = (int*)(partial + (size_t)nblk * (size_t)(n_sets + 1));
    hipLaunchKernelGGL(eval_metrics_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (int)n_rows, n_classes, logits,
                       (long)ld_logits, labels, weights, (long)ld_weights, n_sets, loss_set, out, partial, arrive);
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}
