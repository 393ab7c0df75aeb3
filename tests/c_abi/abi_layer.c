/* The ACM layer operator driven from plain C -- no Python, no torch: the calls that replace
 * GraphConvolution.forward (ACM-Geometric/layers.py:78-116; attention3 at :57-63) and its autograd replay.
 *
 *   literal form        acm_gemm (X [W_L|W_H|W_I]) -> acm_conv_fwd -> acm_conv_bwd_local -> acm_conv_bwd_spmm (on
 *                       acm_csr_transpose) -> acm_gemm (X^T dZ)
 *   aggregate-first     acm_conv_agg_fwd -> acm_conv_agg_bwd, and the pipelined pair (P given, the backward carrying the
 *                       next step's gather over the operator's id streams)
 *
 * on a 300-node ring (+-8 neighbours) with a hub row that is split into several work items, for the ACM-GCN+ layer
 * (ReLU after the filter, LayerNorm in the attention head, sigmoid / 3x3 mix / softmax, scale 3).  Checked against a
 * host restatement of the forward in double precision (plain loops), and the gradients against central finite
 * differences of that host forward (so the only thing restated on the host is the forward the reference defines).
 * Compiled and run by tests/test_gpu_c_abi.py with the same command line as abi_smoke.c.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acm_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define CHECK_ACM(x) do { int s_ = (x); if (s_ != ACM_OK) { printf("acm error %d (%s) at line %d\n", s_, acm_last_error(), __LINE__); return 3; } } while (0)

enum { N = 300, RING = 8, FIN = 6, FPAD = 8, F = 24, K = 3, NW = 3 * FIN * F, NPAR = NW + 3 * K * F + K * K };

static void* to_dev(const void* h, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) return NULL;
    if (h && bytes && hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}
static int from_dev(void* h, const void* d, size_t bytes) { return hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1; }

static unsigned lcg_state = 12345u;
static float rnd(void) {                       /* uniform in [-1, 1) */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)((lcg_state >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}

/* graph (host CSR of A_low = D^-1 (A + I)) */
static int ip[N + 1], *ix;
static float* av;

/* Host forward in double: theta = [W_L | W_H | W_I (FIN x F each)][att_vec K x F][ln_w K x F][ln_b K x F][mix K x K] */
static void host_forward(const double* theta, const float* x, double* out /* N x F */, double* att /* N x K */) {
    const double *W = theta, *v = theta + NW, *lw = v + K * F, *lb = lw + K * F, *mix = lb + K * F;
    static double z[K][N][F], h[K][N][F];
    for (int c = 0; c < K; ++c)
        for (int r = 0; r < N; ++r)
            for (int j = 0; j < F; ++j) {
                double s = 0;
                for (int f = 0; f < FIN; ++f) s += (double)x[r * FIN + f] * W[(c * FIN + f) * F + j];
                z[c][r][j] = s;
            }
    for (int r = 0; r < N; ++r)
        for (int j = 0; j < F; ++j) {
            double pl = 0, ph = 0;
            for (int k = ip[r]; k < ip[r + 1]; ++k) { pl += (double)av[k] * z[0][ix[k]][j]; ph += (double)av[k] * z[1][ix[k]][j]; }
            h[0][r][j] = fmax(pl, 0.0);                         /* relu(A_low Z_L)        layers.py:102 */
            h[1][r][j] = fmax(z[1][r][j] - ph, 0.0);            /* relu((I - A_low) Z_H)  layers.py:103 */
            h[2][r][j] = fmax(z[2][r][j], 0.0);                 /* relu(Z_I)              layers.py:104 */
        }
    for (int r = 0; r < N; ++r) {
        double g[K], logit[K], mx = -1e300, den = 0;
        for (int c = 0; c < K; ++c) {
            double mean = 0, var = 0, dot = 0;
            for (int j = 0; j < F; ++j) mean += h[c][r][j];
            mean /= F;
            for (int j = 0; j < F; ++j) var += (h[c][r][j] - mean) * (h[c][r][j] - mean);
            const double rstd = 1.0 / sqrt(var / F + 1e-5);    /* nn.LayerNorm: biased variance, eps 1e-5 */
            for (int j = 0; j < F; ++j) dot += ((h[c][r][j] - mean) * rstd * lw[c * F + j] + lb[c * F + j]) * v[c * F + j];
            g[c] = 1.0 / (1.0 + exp(-dot));
        }
        for (int j = 0; j < K; ++j) {
            double a = 0;
            for (int c = 0; c < K; ++c) a += g[c] * mix[c * K + j];
            logit[j] = a / K;                                   /* T = 3, layers.py:62 */
            mx = fmax(mx, logit[j]);
        }
        for (int j = 0; j < K; ++j) { logit[j] = exp(logit[j] - mx); den += logit[j]; }
        for (int j = 0; j < K; ++j) att[r * K + j] = logit[j] / den;
        for (int j = 0; j < F; ++j)
            out[r * F + j] = 3.0 * (att[r * K] * h[0][r][j] + att[r * K + 1] * h[1][r][j] + att[r * K + 2] * h[2][r][j]);
    }
}

static double loss_of(const double* theta, const float* x, const float* g0, double* out, double* att) {
    host_forward(theta, x, out, att);
    double l = 0;
    for (int i = 0; i < N * F; ++i) l += out[i] * g0[i];
    return l;
}

static double max_abs(const double* a, int n) { double m = 0; for (int i = 0; i < n; ++i) m = fmax(m, fabs(a[i])); return m; }

int main(void) {
    if (acm_version() != ACM_ABI_VERSION) { printf("ABI version mismatch\n"); return 1; }
    /* ---- graph: hub row 0 + ring of +-RING neighbours, self loop in every row; values 1 / row length */
    ix = (int*)malloc((size_t)N * (2 * RING + 2 + N / N) * sizeof(int) + N * sizeof(int));
    av = (float*)malloc((size_t)N * (2 * RING + 2) * sizeof(float) + N * sizeof(float));
    int nnz = 0;
    for (int r = 0; r < N; ++r) {
        ip[r] = nnz;
        char* mark = (char*)calloc(N, 1);
        if (r == 0) memset(mark, 1, N);
        else {
            mark[0] = mark[r] = 1;
            for (int d = 1; d <= RING; ++d) { mark[(r + d) % N] = 1; mark[(r - d + N) % N] = 1; }
        }
        int len = 0;
        for (int c = 0; c < N; ++c) len += mark[c];
        for (int c = 0; c < N; ++c) if (mark[c]) { ix[nnz] = c; av[nnz++] = 1.0f / (float)len; }
        free(mark);
    }
    ip[N] = nnz;

    /* ---- inputs and parameters */
    static float x[N * FIN], xpad[N * FPAD], g0[N * F], theta_f[NPAR], wcat[FIN * 3 * F];
    static double theta[NPAR];
    for (int i = 0; i < N * FIN; ++i) x[i] = rnd();
    for (int r = 0; r < N; ++r) for (int f = 0; f < FPAD; ++f) xpad[r * FPAD + f] = f < FIN ? x[r * FIN + f] : 0.f;
    for (int i = 0; i < N * F; ++i) g0[i] = rnd();
    for (int i = 0; i < NW; ++i) theta_f[i] = 0.4f * rnd();
    for (int i = 0; i < K * F; ++i) theta_f[NW + i] = rnd();                       /* att_vec  U(-1, 1), layers.py:46-49 */
    for (int i = 0; i < K * F; ++i) theta_f[NW + K * F + i] = 1.0f + 0.3f * rnd(); /* LN gamma */
    for (int i = 0; i < K * F; ++i) theta_f[NW + 2 * K * F + i] = 0.3f * rnd();    /* LN beta  */
    for (int i = 0; i < K * K; ++i) theta_f[NW + 3 * K * F + i] = 0.6f * rnd();    /* att_vec (3 x 3 mix) */
    for (int i = 0; i < NPAR; ++i) theta[i] = theta_f[i];
    for (int c = 0; c < 3; ++c)
        for (int f = 0; f < FIN; ++f)
            for (int j = 0; j < F; ++j) wcat[f * 3 * F + c * F + j] = theta_f[(c * FIN + f) * F + j];

    /* ---- host reference: forward, and gradients of L = sum(out * g0) by central differences */
    static double out_ref[N * F], att_ref[N * K], tmp_out[N * F], tmp_att[N * K], grad_fd[NPAR];
    loss_of(theta, x, g0, out_ref, att_ref);
    for (int i = 0; i < NPAR; ++i) {
        const double keep = theta[i], hstep = 1e-5;
        theta[i] = keep + hstep;
        const double lp = loss_of(theta, x, g0, tmp_out, tmp_att);
        theta[i] = keep - hstep;
        const double lm = loss_of(theta, x, g0, tmp_out, tmp_att);
        theta[i] = keep;
        grad_fd[i] = (lp - lm) / (2 * hstep);
    }

    /* ---- device buffers */
    int* d_ip = (int*)to_dev(ip, sizeof(ip));
    int* d_ix = (int*)to_dev(ix, nnz * sizeof(int));
    float* d_av = (float*)to_dev(av, nnz * sizeof(float));
    float* d_x = (float*)to_dev(x, sizeof(x));
    float* d_xpad = (float*)to_dev(xpad, sizeof(xpad));
    float* d_g0 = (float*)to_dev(g0, sizeof(g0));
    float* d_theta = (float*)to_dev(theta_f, sizeof(theta_f));
    float* d_wcat = (float*)to_dev(wcat, sizeof(wcat));
    float* d_z = (float*)to_dev(NULL, N * 3 * F * sizeof(float));
    float* d_out = (float*)to_dev(NULL, N * F * sizeof(float));
    float* d_pre = (float*)to_dev(NULL, N * 2 * F * sizeof(float));
    float* d_att = (float*)to_dev(NULL, N * 4 * sizeof(float));
    float* d_g = (float*)to_dev(NULL, N * 2 * F * sizeof(float));
    float* d_dz = (float*)to_dev(NULL, N * 3 * F * sizeof(float));
    float* d_dhead = (float*)to_dev(NULL, (3 * K * F + K * K) * sizeof(float));
    float* d_dw = (float*)to_dev(NULL, FIN * 3 * F * sizeof(float));
    if (!d_ip || !d_ix || !d_av || !d_x || !d_xpad || !d_g0 || !d_theta || !d_wcat || !d_z || !d_out || !d_pre || !d_att ||
        !d_g || !d_dz || !d_dhead || !d_dw) { printf("device allocation failed\n"); return 2; }
    const float *d_v = d_theta + NW, *d_lw = d_v + K * F, *d_lb = d_lw + K * F, *d_mix = d_lb + K * F;

    acm_csr_t *a = NULL, *at = NULL;
    CHECK_ACM(acm_csr_create(N, N, nnz, d_ip, d_ix, d_av, 64, &a));       /* chunk 64: the hub row becomes 5 work items */
    CHECK_ACM(acm_csr_transpose(a, 64, &at));
    acm_csr_info_t info;
    CHECK_ACM(acm_csr_info(a, &info));
    if (info.n_long_rows != 1 || info.nnz != nnz) { printf("unexpected handle info\n"); return 4; }
    size_t ws_bytes = 0, ws_t_bytes = 0, gws = 0, k3_bytes = 0;
    CHECK_ACM(acm_spmm_workspace_bytes(a, 2 * F, &ws_bytes));
    CHECK_ACM(acm_spmm_workspace_bytes(at, 2 * F, &ws_t_bytes));
    void* ws = to_dev(NULL, ws_bytes);
    void* ws_t = to_dev(NULL, ws_t_bytes);

    /* ======================= literal form ======================= */
    CHECK_ACM(acm_gemm_workspace_bytes(0, 0, N, 3 * F, FIN, &gws));
    void* gw = to_dev(NULL, gws);
    CHECK_ACM(acm_gemm(0, 0, N, 3 * F, FIN, d_x, FIN, d_wcat, 3 * F, d_z, 3 * F, 0, gw, gws, NULL));   /* torch.mm x3, :87-89 */
    acm_conv_fwd_t p;
    memset(&p, 0, sizeof(p));
    p.f_out = F; p.n_channels = K; p.relu_after = 1; p.relu_mlp = 1; p.layernorm = 1; p.scale = 3.0f;
    p.g_low = d_z; p.ld_g_low = 3 * F;
    p.g_high = d_z + F; p.ld_g_high = 3 * F;
    p.s_high = d_z + F; p.ld_s_high = 3 * F;
    p.s_mlp = d_z + 2 * F; p.ld_s_mlp = 3 * F;
    for (int c = 0; c < K; ++c) { p.att_vec[c] = d_v + c * F; p.ln_weight[c] = d_lw + c * F; p.ln_bias[c] = d_lb + c * F; }
    p.att_mix = d_mix;
    p.out = d_out; p.ld_out = F;
    p.pre = d_pre; p.ld_pre = 2 * F;
    p.att = d_att;
    CHECK_ACM(acm_conv_fwd(a, &p, ws, ws_bytes, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    static float out[N * F], att[N * 4];
    if (from_dev(out, d_out, sizeof(out)) || from_dev(att, d_att, sizeof(att))) return 2;
    double e_out = 0, e_att = 0;
    const double out_scale = fmax(1.0, max_abs(out_ref, N * F));
    for (int i = 0; i < N * F; ++i) e_out = fmax(e_out, fabs(out[i] - out_ref[i]));
    for (int r = 0; r < N; ++r) for (int c = 0; c < K; ++c) e_att = fmax(e_att, fabs(att[r * 4 + c] - att_ref[r * K + c]));
    if (e_out > 2e-5 * out_scale || e_att > 2e-5) { printf("acm_conv_fwd mismatch: out %g att %g\n", e_out, e_att); return 5; }

    /* backward: K3 (row-local) -> K4 (transposed gather) -> X^T dZ */
    acm_conv_bwd_local_t q;
    memset(&q, 0, sizeof(q));
    q.f_out = F; q.n_channels = K; q.relu_after = 1; q.relu_mlp = 1; q.layernorm = 1; q.scale = 3.0f;
    q.grad_out = d_g0; q.ld_grad_out = F;
    q.pre = d_pre; q.ld_pre = 2 * F;
    q.s_mlp = d_z + 2 * F; q.ld_s_mlp = 3 * F;
    for (int c = 0; c < K; ++c) {
        q.att_vec[c] = d_v + c * F; q.ln_weight[c] = d_lw + c * F; q.ln_bias[c] = d_lb + c * F;
        q.d_att_vec[c] = d_dhead + c * F; q.d_ln_weight[c] = d_dhead + (K + c) * F; q.d_ln_bias[c] = d_dhead + (2 * K + c) * F;
    }
    q.att_mix = d_mix;
    q.d_att_mix = d_dhead + 3 * K * F;
    q.g_low = d_g; q.ld_g_low = 2 * F;
    q.g_high = d_g + F; q.ld_g_high = 2 * F;
    q.g_mlp = d_dz + 2 * F; q.ld_g_mlp = 3 * F;
    CHECK_ACM(acm_conv_bwd_local_workspace_bytes(N, F, K, &k3_bytes));
    void* k3w = to_dev(NULL, k3_bytes);
    CHECK_ACM(acm_conv_bwd_local(N, &q, k3w, k3_bytes, NULL));
    acm_conv_bwd_spmm_t r4;
    memset(&r4, 0, sizeof(r4));
    r4.f_out = F;
    r4.g_low = d_g; r4.ld_g_low = 2 * F;
    r4.g_high = d_g + F; r4.ld_g_high = 2 * F;
    r4.s_high = d_g + F; r4.ld_s_high = 2 * F;
    r4.dz_low = d_dz; r4.ld_dz_low = 3 * F;
    r4.dz_high = d_dz + F; r4.ld_dz_high = 3 * F;
    CHECK_ACM(acm_conv_bwd_spmm(at, &r4, ws_t, ws_t_bytes, NULL));
    size_t gws2 = 0;
    CHECK_ACM(acm_gemm_workspace_bytes(1, 0, FIN, 3 * F, N, &gws2));
    void* gw2 = to_dev(NULL, gws2);
    CHECK_ACM(acm_gemm(1, 0, FIN, 3 * F, N, d_x, FIN, d_dz, 3 * F, d_dw, 3 * F, 0, gw2, gws2, NULL));   /* dWcat = X^T dZ */
    CHECK_HIP(hipDeviceSynchronize());
    static float dw[FIN * 3 * F], dhead[3 * K * F + K * K];
    if (from_dev(dw, d_dw, sizeof(dw)) || from_dev(dhead, d_dhead, sizeof(dhead))) return 2;
    const double gw_scale = fmax(1e-3, max_abs(grad_fd, NW)), gh_scale = fmax(1e-3, max_abs(grad_fd + NW, NPAR - NW));
    double e_dw = 0, e_dh = 0;
    for (int c = 0; c < 3; ++c)
        for (int f = 0; f < FIN; ++f)
            for (int j = 0; j < F; ++j)
                e_dw = fmax(e_dw, fabs(dw[f * 3 * F + c * F + j] - grad_fd[(c * FIN + f) * F + j]));
    for (int i = 0; i < 3 * K * F + K * K; ++i) e_dh = fmax(e_dh, fabs(dhead[i] - grad_fd[NW + i]));
    if (e_dw > 2e-3 * gw_scale || e_dh > 2e-3 * gh_scale) {
        printf("literal backward mismatch vs finite differences: dW %g (scale %g) head %g (scale %g)\n", e_dw, gw_scale, e_dh, gh_scale);
        return 6;
    }

    /* ======================= aggregate-first form ======================= */
    float* d_out2 = (float*)to_dev(NULL, N * F * sizeof(float));
    float* d_agg = (float*)to_dev(NULL, N * FPAD * sizeof(float));
    float* d_att2 = (float*)to_dev(NULL, N * 4 * sizeof(float));
    float* d_stats = (float*)to_dev(NULL, N * 4 * K * sizeof(float));
    float* d_dpar = (float*)to_dev(NULL, NPAR * sizeof(float));
    acm_conv_agg_fwd_t u;
    memset(&u, 0, sizeof(u));
    u.f_in = FIN; u.f_pad = FPAD; u.f_out = F; u.relu_after = 1; u.relu_mlp = 1; u.layernorm = 1; u.scale = 3.0f;
    u.xg = d_xpad; u.ld_xg = FPAD;
    u.xs = d_xpad; u.ld_xs = FPAD;
    u.w_low = d_theta; u.w_high = d_theta + FIN * F; u.w_mlp = d_theta + 2 * FIN * F; u.ld_w = F;
    for (int c = 0; c < K; ++c) { u.att_vec[c] = d_v + c * F; u.ln_weight[c] = d_lw + c * F; u.ln_bias[c] = d_lb + c * F; }
    u.att_mix = d_mix;
    u.out = d_out2; u.ld_out = F;
    u.agg = d_agg; u.ld_agg = FPAD;
    u.att = d_att2;
    u.n_channels = K;
    u.head_stats = d_stats; u.ld_head_stats = 4 * K;
    CHECK_ACM(acm_conv_agg_fwd(a, &u, ws, ws_bytes, NULL));
    acm_conv_agg_bwd_t ub;
    memset(&ub, 0, sizeof(ub));
    ub.f_in = FIN; ub.f_pad = FPAD; ub.f_out = F; ub.relu_after = 1; ub.relu_mlp = 1; ub.layernorm = 1; ub.scale = 3.0f;
    ub.grad_out = d_g0; ub.ld_grad_out = F;
    ub.agg = d_agg; ub.ld_agg = FPAD;
    ub.xs = d_xpad; ub.ld_xs = FPAD;
    ub.w_low = u.w_low; ub.w_high = u.w_high; ub.w_mlp = u.w_mlp; ub.ld_w = F;
    for (int c = 0; c < K; ++c) { ub.att_vec[c] = u.att_vec[c]; ub.ln_weight[c] = u.ln_weight[c]; ub.ln_bias[c] = u.ln_bias[c]; }
    ub.att_mix = d_mix;
    ub.d_params = d_dpar;
    ub.n_channels = K;
    ub.head_stats = d_stats; ub.ld_head_stats = 4 * K;
    size_t ab_bytes = 0;
    CHECK_ACM(acm_conv_agg_bwd_workspace_bytes(N, FIN, F, &ab_bytes));
    void* abw = to_dev(NULL, ab_bytes);
    CHECK_ACM(acm_conv_agg_bwd(N, &ub, abw, ab_bytes, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    static float out2[N * F], agg[N * FPAD], dpar[NPAR];
    if (from_dev(out2, d_out2, sizeof(out2)) || from_dev(agg, d_agg, sizeof(agg)) || from_dev(dpar, d_dpar, sizeof(dpar))) return 2;
    double e_out2 = 0, e_agg = 0, e_par_w = 0, e_par_h = 0;
    for (int i = 0; i < N * F; ++i) e_out2 = fmax(e_out2, fabs(out2[i] - out_ref[i]));
    for (int r = 0; r < N; ++r)
        for (int f = 0; f < FIN; ++f) {
            double s = 0;
            for (int k = ip[r]; k < ip[r + 1]; ++k) s += (double)av[k] * x[ix[k] * FIN + f];
            e_agg = fmax(e_agg, fabs(s - agg[r * FPAD + f]));
        }
    for (int i = 0; i < NW; ++i) e_par_w = fmax(e_par_w, fabs(dpar[i] - grad_fd[i]));
    for (int i = NW; i < NPAR; ++i) e_par_h = fmax(e_par_h, fabs(dpar[i] - grad_fd[i]));
    if (e_out2 > 2e-5 * out_scale || e_agg > 1e-5) { printf("acm_conv_agg_fwd mismatch: out %g agg %g\n", e_out2, e_agg); return 7; }
    if (e_par_w > 2e-3 * gw_scale || e_par_h > 2e-3 * gh_scale) {
        printf("acm_conv_agg_bwd mismatch vs finite differences: dW %g (scale %g) head %g (scale %g)\n", e_par_w, gw_scale, e_par_h, gh_scale);
        return 8;
    }

    /* ======================= the NEXT step's input aggregation carried by the backward =======================
     * (acm_conv_agg_bwd_t.next_agg, acm_conv_agg_fwd_t.agg_given / agg_copy / xs_copy; width 64, pattern-only operator
     * with id streams).  Checked: the carried product against host loops, the copies bit for bit, the gradients against
     * the same call without the gather. */
    enum { F2 = 64, NW2 = 3 * FIN * F2, NPAR2 = NW2 + 3 * K * F2 + K * K };
    static float theta2[NPAR2], g2[N * F2], rs[N], xnext[N * FPAD];
    for (int i = 0; i < NW2; ++i) theta2[i] = 0.4f * rnd();
    for (int i = 0; i < K * F2; ++i) theta2[NW2 + i] = rnd();
    for (int i = 0; i < K * F2; ++i) theta2[NW2 + K * F2 + i] = 1.0f + 0.3f * rnd();
    for (int i = 0; i < K * F2; ++i) theta2[NW2 + 2 * K * F2 + i] = 0.3f * rnd();
    for (int i = 0; i < K * K; ++i) theta2[NW2 + 3 * K * F2 + i] = 0.6f * rnd();
    for (int i = 0; i < N * F2; ++i) g2[i] = rnd();
    for (int r = 0; r < N; ++r) rs[r] = 1.0f / (float)(ip[r + 1] - ip[r]);
    for (int r = 0; r < N; ++r) for (int f = 0; f < FPAD; ++f) xnext[r * FPAD + f] = f < FIN ? rnd() : 0.f;
    float* d_theta2 = (float*)to_dev(theta2, sizeof(theta2));
    float* d_g2 = (float*)to_dev(g2, sizeof(g2));
    float* d_rs = (float*)to_dev(rs, sizeof(rs));
    float* d_xnext = (float*)to_dev(xnext, sizeof(xnext));
    float* d_o64 = (float*)to_dev(NULL, N * F2 * sizeof(float));
    float* d_o64b = (float*)to_dev(NULL, N * F2 * sizeof(float));
    float* d_p = (float*)to_dev(NULL, N * FPAD * sizeof(float));
    float* d_pc = (float*)to_dev(NULL, N * FPAD * sizeof(float));
    float* d_xc = (float*)to_dev(NULL, N * FPAD * sizeof(float));
    float* d_pnext = (float*)to_dev(NULL, N * FPAD * sizeof(float));
    float* d_st64 = (float*)to_dev(NULL, N * 4 * K * sizeof(float));
    float* d_dpa = (float*)to_dev(NULL, NPAR2 * sizeof(float));
    float* d_dpb = (float*)to_dev(NULL, NPAR2 * sizeof(float));
    if (!d_theta2 || !d_g2 || !d_rs || !d_xnext || !d_o64 || !d_o64b || !d_p || !d_pc || !d_xc || !d_pnext || !d_st64 || !d_dpa || !d_dpb) return 2;
    acm_csr_t* ap = NULL;                                   /* the same graph as a pattern: values = rs[row] */
    CHECK_ACM(acm_csr_create(N, N, nnz, d_ip, d_ix, NULL, 64, &ap));
    CHECK_ACM(acm_csr_build_streams(ap, 4 * ((N + 15) / 16), 64));          /* 64 neighbours per piece: the hub row has five */
    acm_csr_info_t sinfo;
    CHECK_ACM(acm_csr_info(ap, &sinfo));
    if (sinfo.stream_waves % 4 != 0 || sinfo.stream_long_rows < 1) { printf("streams: %d waves, %d long rows\n", sinfo.stream_waves, sinfo.stream_long_rows); return 11; }
    acm_conv_agg_fwd_t w = u;
    w.f_out = F2; w.ld_w = F2;
    w.w_low = d_theta2; w.w_high = d_theta2 + FIN * F2; w.w_mlp = d_theta2 + 2 * FIN * F2;
    for (int c = 0; c < K; ++c) { w.att_vec[c] = d_theta2 + NW2 + c * F2; w.ln_weight[c] = d_theta2 + NW2 + (K + c) * F2; w.ln_bias[c] = d_theta2 + NW2 + (2 * K + c) * F2; }
    w.att_mix = d_theta2 + NW2 + 3 * K * F2;
    w.row_scale = d_rs;
    w.out = d_o64; w.ld_out = F2;
    w.agg = d_p;
    w.head_stats = d_st64;
    w.post_relu = 1;                                        /* the hidden layer's fused ReLU: the backward reads `out` as its mask */
    size_t wsp_bytes = 0;
    CHECK_ACM(acm_spmm_workspace_bytes(ap, F2, &wsp_bytes));
    void* wsp = to_dev(NULL, wsp_bytes);
    CHECK_ACM(acm_conv_agg_fwd(ap, &w, wsp, wsp_bytes, NULL));             /* step t, plain: gathers P itself */
    acm_conv_agg_bwd_t wb = ub;
    wb.f_out = F2; wb.ld_w = F2; wb.ld_grad_out = F2; wb.grad_out = d_g2;
    wb.w_low = w.w_low; wb.w_high = w.w_high; wb.w_mlp = w.w_mlp;
    for (int c = 0; c < K; ++c) { wb.att_vec[c] = w.att_vec[c]; wb.ln_weight[c] = w.ln_weight[c]; wb.ln_bias[c] = w.ln_bias[c]; }
    wb.att_mix = w.att_mix;
    wb.agg = d_p; wb.head_stats = d_st64; wb.d_params = d_dpa;
    wb.post_relu = 1; wb.out = d_o64; wb.ld_out = F2;
    size_t ab2_bytes = 0;
    CHECK_ACM(acm_conv_agg_bwd_workspace_bytes(N, FIN, F2, &ab2_bytes));
    void* abw2 = to_dev(NULL, ab2_bytes);
    CHECK_ACM(acm_conv_agg_bwd(N, &wb, abw2, ab2_bytes, NULL));
    /* the same step as a pipelined loop runs it: P given, copies left for the backward, which carries A_low x_next */
    acm_conv_agg_fwd_t wp = w;
    wp.agg_given = 1; wp.out = d_o64b;
    wp.agg_copy = d_pc; wp.ld_agg_copy = FPAD; wp.xs_copy = d_xc; wp.ld_xs_copy = FPAD;
    /* ... and (ABI 22) the row-local stage refills `xs` in place with the next step's input (next_x; no dropout here: p = 0),
     * so the table the backward's gather waves walk IS `xs` */
    wp.next_x = d_xnext; wp.ld_next_x = FPAD; wp.next_drop.p = 0.f;
    CHECK_ACM(acm_conv_agg_fwd(ap, &wp, wsp, wsp_bytes, NULL));
    acm_conv_agg_bwd_t wq = wb;
    wq.agg = d_pc; wq.xs = d_xc; wq.d_params = d_dpb; wq.out = d_o64b;
    wq.next_a = ap; wq.next_xg = wp.xs; wq.ld_next_xg = FPAD; wq.next_row_scale = d_rs; wq.next_agg = d_pnext; wq.ld_next_agg = FPAD;
    CHECK_ACM(acm_conv_agg_bwd(N, &wq, abw2, ab2_bytes, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    static float o64[N * F2], o64b[N * F2], pp[N * FPAD], pc[N * FPAD], xc[N * FPAD], pnext[N * FPAD], dpa[NPAR2], dpb[NPAR2];
    if (from_dev(o64, d_o64, sizeof(o64)) || from_dev(o64b, d_o64b, sizeof(o64b)) || from_dev(pp, d_p, sizeof(pp)) || from_dev(pc, d_pc, sizeof(pc)) ||
        from_dev(xc, d_xc, sizeof(xc)) || from_dev(pnext, d_pnext, sizeof(pnext)) || from_dev(dpa, d_dpa, sizeof(dpa)) || from_dev(dpb, d_dpb, sizeof(dpb))) return 2;
    if (memcmp(pp, pc, sizeof(pp)) || memcmp(xpad, xc, sizeof(xc))) { printf("agg_copy / xs_copy differ from agg / xs\n"); return 12; }
    double e_o = 0, s_o = 0, e_pn = 0, e_dp = 0, s_dp = 0;
    for (int i = 0; i < N * F2; ++i) { e_o = fmax(e_o, fabs(o64[i] - o64b[i])); s_o = fmax(s_o, fabs(o64[i])); }
    for (int r = 0; r < N; ++r)
        for (int f = 0; f < FPAD; ++f) {
            double sum = 0;
            for (int k = ip[r]; k < ip[r + 1]; ++k) sum += xnext[ix[k] * FPAD + f];
            e_pn = fmax(e_pn, fabs(sum * rs[r] - pnext[r * FPAD + f]));
        }
    for (int i = 0; i < NPAR2; ++i) { e_dp = fmax(e_dp, fabs(dpa[i] - dpb[i])); s_dp = fmax(s_dp, fabs(dpa[i])); }
    if (e_o > 2e-5 * s_o || e_pn > 1e-5 || e_dp > 2e-5 * s_dp) { printf("carried gather: out %g (scale %g) next P %g d_params %g (scale %g)\n", e_o, s_o, e_pn, e_dp, s_dp); return 13; }
    acm_conv_agg_bwd_t bad3 = wq;
    bad3.next_agg = d_pc;                                   /* aliases the backward's own operand */
    if (acm_conv_agg_bwd(N, &bad3, abw2, ab2_bytes, NULL) != ACM_EINVAL) { printf("aliasing next_agg accepted\n"); return 14; }
    acm_csr_destroy(ap);

    /* errors are codes, not crashes */
    acm_conv_fwd_t bad = p;
    bad.out = NULL;
    if (acm_conv_fwd(a, &bad, ws, ws_bytes, NULL) != ACM_EINVAL) { printf("NULL output accepted\n"); return 9; }
    acm_conv_agg_fwd_t bad2 = u;
    bad2.f_pad = 4;
    if (acm_conv_agg_fwd(a, &bad2, ws, ws_bytes, NULL) != ACM_ESHAPE) { printf("wrong f_pad accepted\n"); return 10; }
    acm_csr_destroy(a);
    acm_csr_destroy(at);
    printf("abi_layer ok: n=%d nnz=%d | fwd max|err| %.2e (literal) %.2e (aggregate-first) | grads vs FD: dW %.2e / %.2e, head %.2e / %.2e (ranges %.2g, %.2g) | carried gather: next P %.2e, d_params %.2e of %.2g\n",
           N, nnz, e_out, e_out2, e_dw, e_par_w, e_dh, e_par_h, gw_scale, gh_scale, e_pn, e_dp, s_dp);
    return 0;
}
