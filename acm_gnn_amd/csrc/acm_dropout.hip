// Counter-based dropout as a stand-alone pass (gfx950): the input-feature dropout of the model
// (ACM-Geometric/models.py:54) written straight into the zero-padded row layout the aggregate-first gather
// wants, so that F.dropout + pad (a fill and a copy) is one launch.  The mask function is acm_drop1
// (acm_common.h) -- the same one the fused layer kernels evaluate in registers for the hidden dropout.
#include "acm_common.h"

namespace {

__global__ __launch_bounds__(256) void dropout_kernel(long n_rows, int n_cols, const float* __restrict__ src, long ld_src,
                                                      float* __restrict__ dst, long ld_dst, int dst_cols,
                                                      acm_dropout_t d) {
    const AcmDropCtx dc = acm_drop_ctx(d);
    const long total = n_rows * dst_cols;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long r = q / dst_cols;
        const int c = (int)(q - r * dst_cols);
        dst[r * ld_dst + c] = c < n_cols ? src[r * ld_src + c] * acm_drop1(dc, r, c) : 0.f;
    }
}

// Wide matrices (>= 64 columns): one thread per (row, Philox block) = the four columns c, c + 16, c + 32, c + 48 of a
// 64-column panel, so every Philox call serves four elements (the per-element form wastes three of its four words).
__global__ __launch_bounds__(256) void dropout_wide_kernel(long n_rows, int n_cols, const float* __restrict__ src,
                                                           long ld_src, float* __restrict__ dst, long ld_dst, int dst_cols,
                                                           acm_dropout_t d) {
    const AcmDropCtx dc = acm_drop_ctx(d);
    // Philox blocks of a row that hold a column at all: 16 per full panel, min(16, columns) of the last one (68 columns -- pokec's
    // 65 padded -- are 20 blocks, not 32)
    const int last = dst_cols - 64 * ((dst_cols - 1) / 64);
    const int bpr = 16 * ((dst_cols - 1) / 64) + (last < 16 ? last : 16);
    const long total = n_rows * bpr;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long r = q / bpr;
        const int b = (int)(q - r * bpr);                     // block id: (col & 15) + 16 * (col >> 6)
        const int c0 = (b & 15) + 64 * (b >> 4);
        float f[4];
        acm_drop4(dc, r, b, f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + 16 * i;
            if (c < dst_cols) dst[r * ld_dst + c] = c < n_cols ? src[r * ld_src + c] * f[i] : 0.f;
        }
    }
}

}  // namespace

extern "C" int acm_dropout(int64_t n_rows, int64_t n_cols, const float* src, int64_t ld_src, float* dst, int64_t ld_dst,
                           int64_t dst_cols, const acm_dropout_t* d, acm_stream_t stream) {
    ACM_REQUIRE(src && dst && d, ACM_EINVAL, "acm_dropout: NULL argument");
    ACM_REQUIRE(n_rows >= 0 && n_cols >= 0 && dst_cols >= n_cols && dst_cols < INT32_MAX && ld_src >= n_cols &&
                    ld_dst >= dst_cols, ACM_ESHAPE, "acm_dropout: bad sizes");
    ACM_REQUIRE(d->p >= 0.f && d->p < 1.f, ACM_EINVAL, "acm_dropout: p = %g outside [0, 1)", (double)d->p);
    ACM_REQUIRE(d->p == 0.f || d->step, ACM_EINVAL, "acm_dropout: step counter is NULL");
    ACM_REQUIRE(n_cols <= 65536, ACM_EUNSUPPORTED, "acm_dropout: more than 65536 columns");
    if (n_rows == 0 || dst_cols == 0) return ACM_OK;
    if (dst_cols >= 64) {
        const long last = dst_cols - 64 * ((dst_cols - 1) / 64);
        long blocks = (n_rows * (16 * ((dst_cols - 1) / 64) + (last < 16 ? last : 16)) + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(dropout_wide_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long)n_rows,
                           (int)n_cols, src, (long)ld_src, dst, (long)ld_dst, (int)dst_cols, *d);
    } else {
        long blocks = (n_rows * dst_cols + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long)n_rows,
                           (int)n_cols, src, (long)ld_src, dst, (long)ld_dst, (int)dst_cols, *d);
    }
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}
