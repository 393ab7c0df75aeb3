/* A plain-C client of libacm_hip.so: no Python, no torch -- only the HIP runtime for device memory.
 * Builds a small graph, runs acm_spmm / acm_spmm_ex (pattern-only handle + row scale) / acm_gemm / acm_adam_step /
 * acm_nll_loss (immediate and deferred + acm_reduce_flush) / acm_eval_metrics
 * and checks them against loops on the host.  Compiled and run by tests/test_gpu_c_abi.py:
 *     gcc -std=c11 -D__HIP_PLATFORM_AMD__ abi_smoke.c -I include -I/opt/rocm/include -L acm_gnn_amd/lib -lacm_hip \
 *         -L/opt/rocm/lib -lamdhip64 -lm -o abi_smoke
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "acm_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define CHECK_ACM(x) do { int s_ = (x); if (s_ != ACM_OK) { printf("acm error %d (%s) at line %d\n", s_, acm_last_error(), __LINE__); return 3; } } while (0)

static void* to_dev(const void* h, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) return NULL;
    if (h && bytes && hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(void) {
    if (acm_version() != ACM_ABI_VERSION) { printf("ABI version mismatch\n"); return 1; }
    enum { N = 300, W = 8 };
    /* ring + a hub: row 0 is connected to everybody, row i to i-1, i, i+1 */
    int* ip = (int*)malloc((N + 1) * sizeof(int));
    int* ix = (int*)malloc((size_t)(N + 4 * N) * sizeof(int));
    float* v = (float*)malloc((size_t)(N + 4 * N) * sizeof(float));
    int nnz = 0;
    for (int r = 0; r < N; ++r) {
        ip[r] = nnz;
        if (r == 0) { for (int c = 0; c < N; ++c) { ix[nnz] = c; v[nnz++] = 1.0f / N; } continue; }
        int cols[4] = {0, r - 1, r, (r + 1) % N}, k = 0, used[4];
        for (int t = 0; t < 4; ++t) {                                   /* sorted, unique */
            int c = cols[t], dup = 0;
            for (int u = 0; u < k; ++u) dup |= used[u] == c;
            if (!dup) used[k++] = c;
        }
        for (int a = 0; a < k; ++a) for (int b = a + 1; b < k; ++b) if (used[b] < used[a]) { int t = used[a]; used[a] = used[b]; used[b] = t; }
        for (int t = 0; t < k; ++t) { ix[nnz] = used[t]; v[nnz++] = 1.0f / k; }
    }
    ip[N] = nnz;
    float* x = (float*)malloc(N * W * sizeof(float));
    for (int i = 0; i < N * W; ++i) x[i] = (float)((i * 37) % 19) / 7.0f - 1.0f;
    int* d_ip = (int*)to_dev(ip, (N + 1) * sizeof(int));
    int* d_ix = (int*)to_dev(ix, nnz * sizeof(int));
    float* d_v = (float*)to_dev(v, nnz * sizeof(float));
    float* d_x = (float*)to_dev(x, N * W * sizeof(float));
    float* d_y = (float*)to_dev(NULL, N * W * sizeof(float));
    if (!d_ip || !d_ix || !d_v || !d_x || !d_y) { printf("device allocation failed\n"); return 2; }

    acm_csr_t *a = NULL, *pat = NULL;
    CHECK_ACM(acm_csr_create(N, N, nnz, d_ip, d_ix, d_v, 64, &a));          /* chunk 64: row 0 is split into 5 items */
    CHECK_ACM(acm_csr_create(N, N, nnz, d_ip, d_ix, NULL, 64, &pat));       /* pattern-only twin */
    acm_csr_info_t info;
    CHECK_ACM(acm_csr_info(a, &info));
    if (info.nnz != nnz || info.n_long_rows != 1) { printf("unexpected handle info\n"); return 4; }
    size_t ws_bytes = 0;
    CHECK_ACM(acm_spmm_workspace_bytes(a, W, &ws_bytes));
    void* ws = to_dev(NULL, ws_bytes);

    /* 1. Y = A X */
    CHECK_ACM(acm_spmm(a, d_x, W, W, d_y, W, ws, ws_bytes, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float* y = (float*)malloc(N * W * sizeof(float));
    CHECK_HIP(hipMemcpy(y, d_y, N * W * sizeof(float), hipMemcpyDeviceToHost));
    double worst = 0;
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < W; ++c) {
            double s = 0;
            for (int k = ip[r]; k < ip[r + 1]; ++k) s += (double)v[k] * x[ix[k] * W + c];
            worst = fmax(worst, fabs(s - y[r * W + c]));
        }
    if (worst > 1e-5) { printf("acm_spmm mismatch %g\n", worst); return 5; }

    /* 2. the same product from the pattern-only handle: relu(row_scale * (P X)) */
    float* rs = (float*)malloc(N * sizeof(float));
    for (int r = 0; r < N; ++r) rs[r] = v[ip[r]];
    float* d_rs = (float*)to_dev(rs, N * sizeof(float));
    acm_spmm_opts_t o = {NULL, d_rs, NULL, 0, NULL, 1, 0};
    CHECK_ACM(acm_spmm_ex(pat, d_x, W, W, d_y, W, &o, ws, ws_bytes, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float* y2 = (float*)malloc(N * W * sizeof(float));
    CHECK_HIP(hipMemcpy(y2, d_y, N * W * sizeof(float), hipMemcpyDeviceToHost));
    for (int i = 0; i < N * W; ++i)
        if (fabs(y2[i] - fmaxf(y[i], 0.f)) > 1e-5) { printf("acm_spmm_ex mismatch at %d: %g vs %g\n", i, y2[i], y[i]); return 6; }

    /* 3. C = X^T X (8 x 8) on the MFMA pipe, split-K workspace from the library */
    float* d_c = (float*)to_dev(NULL, W * W * sizeof(float));
    size_t gws = 0;
    CHECK_ACM(acm_gemm_workspace_bytes(1, 0, W, W, N, &gws));
    void* gw = to_dev(NULL, gws);
    CHECK_ACM(acm_gemm(1, 0, W, W, N, d_x, W, d_x, W, d_c, W, 0, gw, gws, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float c[W * W];
    CHECK_HIP(hipMemcpy(c, d_c, sizeof(c), hipMemcpyDeviceToHost));
    for (int i = 0; i < W; ++i)
        for (int j = 0; j < W; ++j) {
            double s = 0;
            for (int r = 0; r < N; ++r) s += (double)x[r * W + i] * x[r * W + j];
            if (fabs(s - c[i * W + j]) > 1e-3) { printf("acm_gemm mismatch (%d,%d): %g vs %g\n", i, j, c[i * W + j], s); return 7; }
        }

    /* 4. one AdamW step on the 64-element matrix */
    float g[W * W], m[W * W] = {0}, vv[W * W] = {0}, step = 0.f, p0[W * W];
    for (int i = 0; i < W * W; ++i) { g[i] = 0.01f * (float)(i - 30); p0[i] = c[i]; }
    acm_adam_tensor_t t = {d_c, (float*)to_dev(g, sizeof(g)), (float*)to_dev(m, sizeof(m)), (float*)to_dev(vv, sizeof(vv)),
                           (float*)to_dev(&step, sizeof(step)), W * W};
    acm_adam_config_t cfg = {0.1, 0.9, 0.999, 1e-8, 0.01, 1, NULL, NULL};
    CHECK_ACM(acm_adam_step(1, &t, &cfg, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(c, d_c, sizeof(c), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&step, t.step, sizeof(step), hipMemcpyDeviceToHost));
    if (step != 1.0f) { printf("step counter %g\n", step); return 8; }
    for (int i = 0; i < W * W; ++i) {
        double p = (double)p0[i] * (1.0 - 0.1 * 0.01);
        double mm = 0.1 * g[i], v2 = 0.001 * (double)g[i] * g[i];
        double upd = (0.1 / (1.0 - 0.9)) * mm / (sqrt(v2) / sqrt(1.0 - 0.999) + 1e-8);
        if (fabs(p - upd - c[i]) > 1e-4 * fmax(1.0, fabs(p))) { printf("acm_adam_step mismatch at %d: %g vs %g\n", i, c[i], p - upd); return 9; }
    }

    /* 5. masked NLL, immediate and with its final sum deferred to acm_reduce_flush (bit-identical) */
    enum { C2 = 2 };
    long long lab[N];
    float wt[N], zz[N * C2];
    for (int i = 0; i < N; ++i) { lab[i] = i % C2; wt[i] = (i % 3 == 0) ? 3.0f / N : 0.f; zz[2 * i] = x[i * W]; zz[2 * i + 1] = x[i * W + 1]; }
    float* d_z = (float*)to_dev(zz, sizeof(zz));
    float* d_dz = (float*)to_dev(NULL, sizeof(zz));
    float* d_w = (float*)to_dev(wt, sizeof(wt));
    long long* d_lab = (long long*)to_dev(lab, sizeof(lab));
    float* d_loss = (float*)to_dev(NULL, 2 * sizeof(float));
    size_t lws = 0;
    CHECK_ACM(acm_nll_loss_workspace_bytes(N, &lws));
    void* lw = to_dev(NULL, lws);
    CHECK_ACM(acm_nll_loss(N, C2, d_z, C2, (const int64_t*)d_lab, d_w, d_loss, d_dz, C2, lw, lws, NULL, NULL));
    acm_reduce_seg_t segs[4];
    acm_reduce_list_t pending = {0, 4, segs};
    void* lw2 = to_dev(NULL, lws);
    CHECK_ACM(acm_nll_loss(N, C2, d_z, C2, (const int64_t*)d_lab, d_w, d_loss + 1, d_dz, C2, lw2, lws, &pending, NULL));
    if (pending.n != 1) { printf("deferred acm_nll_loss appended %d segments\n", pending.n); return 12; }
    CHECK_ACM(acm_reduce_flush(&pending, NULL));
    if (pending.n != 0) { printf("acm_reduce_flush left %d segments\n", pending.n); return 13; }
    CHECK_HIP(hipDeviceSynchronize());
    float loss2[2];
    CHECK_HIP(hipMemcpy(loss2, d_loss, sizeof(loss2), hipMemcpyDeviceToHost));
    double want = 0;
    for (int i = 0; i < N; ++i) {
        double a0 = zz[2 * i], a1 = zz[2 * i + 1], mx = fmax(a0, a1), lse = mx + log(exp(a0 - mx) + exp(a1 - mx));
        want += wt[i] * (lse - zz[2 * i + lab[i]]);
    }
    if (loss2[0] != loss2[1] || fabs(loss2[0] - want) > 1e-5) { printf("acm_nll_loss %g / deferred %g vs %g\n", loss2[0], loss2[1], want); return 14; }
    pending.cap = 0;
    if (acm_nll_loss(N, C2, d_z, C2, (const int64_t*)d_lab, d_w, d_loss, d_dz, C2, lw, lws, &pending, NULL) != ACM_ENOMEM) { printf("full deferral list accepted\n"); return 15; }

    /* 5b. acm_eval_metrics (ABI 28): accuracy on two index sets + the NLL on the second, one launch, twice (the arrival
     *     counter in the workspace resets itself) */
    {
        float w2[2 * N];
        int n0 = 0, n1 = 0;
        for (int i = 0; i < N; ++i) { n0 += i % 3 == 0; n1 += i % 3 == 1; }
        for (int i = 0; i < N; ++i) { w2[i] = (i % 3 == 0) ? 1.0f / n0 : 0.f; w2[N + i] = (i % 3 == 1) ? 1.0f / n1 : 0.f; }
        long long lab2[N];
        for (int i = 0; i < N; ++i) lab2[i] = (i % 3 == 2) ? -1 : lab[i];              /* rows outside both sets: unlabeled */
        float* d_w2 = (float*)to_dev(w2, sizeof(w2));
        long long* d_lab2 = (long long*)to_dev(lab2, sizeof(lab2));
        float* d_m = (float*)to_dev(NULL, 3 * sizeof(float));
        size_t mws = 0;
        CHECK_ACM(acm_eval_metrics_workspace_bytes(N, 2, &mws));
        void* mw = to_dev(NULL, mws);
        CHECK_HIP(hipMemset(mw, 0, mws));
        double acc0 = 0, acc1 = 0, nll1 = 0;
        for (int i = 0; i < N; ++i) {
            if (i % 3 == 2) continue;
            double a0 = zz[2 * i], a1 = zz[2 * i + 1], mx = fmax(a0, a1), lse = mx + log(exp(a0 - mx) + exp(a1 - mx));
            int arg = a1 > a0 ? 1 : 0;                                                  /* the first maximum wins */
            if (i % 3 == 0) acc0 += (arg == lab[i]) ? 1.0 / n0 : 0.0;
            else { acc1 += (arg == lab[i]) ? 1.0 / n1 : 0.0; nll1 += (lse - zz[2 * i + lab[i]]) / n1; }
        }
        for (int rep = 0; rep < 2; ++rep) {
            float m3[3];
            CHECK_ACM(acm_eval_metrics(N, C2, d_z, C2, (const int64_t*)d_lab2, d_w2, N, 2, 1, d_m, mw, mws, NULL));
            CHECK_HIP(hipDeviceSynchronize());
            CHECK_HIP(hipMemcpy(m3, d_m, sizeof(m3), hipMemcpyDeviceToHost));
            if (fabs(m3[0] - acc0) > 1e-6 || fabs(m3[1] - acc1) > 1e-6 || fabs(m3[2] - nll1) > 1e-5) {
                printf("acm_eval_metrics (%d) %g %g %g vs %g %g %g\n", rep, m3[0], m3[1], m3[2], acc0, acc1, nll1);
                return 16;
            }
        }
        if (acm_eval_metrics(N, C2, d_z, C2, (const int64_t*)d_lab2, d_w2, N, 2, 2, d_m, mw, mws, NULL) != ACM_ESHAPE) { printf("bad loss set accepted\n"); return 17; }
    }

    /* 6. errors are codes + messages, never crashes */
    if (acm_spmm(NULL, d_x, W, W, d_y, W, ws, ws_bytes, NULL) != ACM_EINVAL) { printf("NULL handle accepted\n"); return 10; }
    if (acm_spmm(a, d_x, W, W, d_y, W, NULL, 0, NULL) != ACM_ENOMEM) { printf("missing workspace accepted\n"); return 11; }
    acm_csr_destroy(a);
    acm_csr_destroy(pat);
    printf("abi_smoke ok: n=%d nnz=%d max|err|=%.2e\n", N, nnz, worst);
    return 0;
}
