// Row-panel fp32 GEMMs for the wide dense projections of the ACM layer on gfx950 (v_mfma_f32_16x16x4_f32: exact fp32, a
// k-ordered fmaf chain), with the counter-based INPUT dropout (ACM-Geometric/models.py:54) applied while the X tile is
// staged -- the dropped copy of X (173 MB written + read on the arXiv-year-shaped graph) never exists:
//
//   NN   Z  = drop(X) W          X: [n, K]   W: [K, N]    n ~ 1e5, K <= 1024, N <= 192      layers.py:86-88,101-103
//   TN   dW = drop(X)^T dZ       dW: [K, N]  dZ: [n, N]   K <= 128,  N <= 192, split over n  (MmBackward of the same)
//
// The older tile kernel (acm_gemm.hip: 64 x 64 tiles, dword loads, 33-float LDS rows) re-reads X once per 64-column
// tile and spends as long on LDS conflicts and guards as on the matrix pipe: 64 TF / 18 % of the HBM stream on these
// shapes.  Here a workgroup owns 128 rows x ALL N columns (NN) or a span of rows x the whole K x N output (TN), so X and dZ
// are read exactly once; tiles sit in LDS with the contraction index outermost and a row stride of 16 mod 32 floats, so
// that the MFMA operand reads of both 16-lane row groups of a half-wave hit 32 distinct banks.
//
// Dropout in the tile load: the keep decision of X[row][col] is word (col >> 4) & 3 of Philox(row, (col & 15) + 16 (col >> 6)),
// so a thread stages the FOUR columns c, c + 16, c + 32, c + 48 of one row of a 64-column slab from one Philox call.
#include "acm_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int RBK = 64;            // contraction slab
constexpr int RBM = 64;            // rows per workgroup (NN): 73 KB of LDS at N = 192, two workgroups per CU -- one stages while
                                   // the other feeds the matrix pipe

__device__ __forceinline__ int pad16(int n) { return ((n + 31) / 32) * 32 + 16; }   // row stride = 16 mod 32 floats

// The contraction loop over KS steps of four: the NEXT step's operands are read from LDS (1 + NT dwords per lane) before the
// current step's MFMAs are issued -- two register sets, so the reads never wait on the matrix pipe and the MFMAs never wait on
// an LDS round trip (hipcc otherwise reuses one register pair: read, wait, two MFMAs, read, wait, ...).
template <int NT, int NA>
__device__ __forceinline__ void mma_steps(const float* __restrict__ As, int sa, int a_off, const float* __restrict__ Bs, int sb,
                                          int m, int g, int ksteps, f32x4 (&acc)[NA][NT]) {
    float a0[NA], b0[NT], a1[NA], b1[NT];
    auto rd = [&](int ks, float (&a)[NA], float (&b)[NT]) {
        const int kk = 4 * ks + g;
#pragma unroll
        for (int i = 0; i < NA; ++i) a[i] = As[kk * sa + a_off + 16 * i + m];
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = Bs[kk * sb + 16 * j + m];
    };
    auto mm = [&](const float (&a)[NA], const float (&b)[NT]) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    rd(0, a0, b0);
    for (int ks = 0; ks < ksteps; ks += 2) {       // ksteps is even (slabs of 64 columns)
        rd(ks + 1, a1, b1);
        mm(a0, b0);
        if (ks + 2 < ksteps) rd(ks + 2, a0, b0);
        mm(a1, b1);
    }
}

// ---- NN: C[m0 .. m0+64, 0 .. N) = drop(A)[.., K] B[K, N];  NT = N / 16 column tiles (<= 12), wave w: rows 16 w .. 16 w + 15
template <int NT>
__global__ __launch_bounds__(256) void gemm_rows_nn_kernel(int M, int N, int K, const float* __restrict__ A, long lda,
                                                           const float* __restrict__ B, long ldb, float* __restrict__ C, long ldc,
                                                           int relu, acm_dropout_t drop, int vecb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int SA = RBM + 16;                   // As[k][row]: 80
    const int SB = pad16(NT * 16);             // Bs[k][col]
    float* As = lds;                           // RBK x SA
    float* Bs = lds + RBK * SA;                // RBK x SB
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * RBM;
    const AcmDropCtx dc = acm_drop_ctx(drop);
    f32x4 acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // staging: thread t handles pairs (row = t / 16 + 16 i, c = t % 16), i = 0..3: columns k0 + c + 16 q, q = 0..3
    const int sc = threadIdx.x & 15, sr = threadIdx.x >> 4;
    for (int k0 = 0; k0 < K; k0 += RBK) {
        if (k0 > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < RBM / 16; ++i) {
            const int r = sr + 16 * i, row = m0 + r;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (row < M) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = k0 + sc + 16 * q;
                    v[q] = col < K ? A[(long)row * lda + col] : 0.f;
                }
                if (dc.on) {
                    // columns k0 + sc + 16 q share block (sc + 16 (k0 >> 6)); word q  (k0 is a multiple of 64)
                    unsigned w[4];
                    acm_philox7(dc, row, sc + 16 * (k0 >> 6), w);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (w[q] >= dc.thresh) ? v[q] * dc.inv_keep : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) As[(sc + 16 * q) * SA + r] = v[q];
        }
        // B slab: RBK x N, float4 along N
        for (int idx = threadIdx.x; idx < RBK * NT * 4; idx += 256) {
            const int kk = idx / (NT * 4), c4 = idx % (NT * 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k0 + kk < K && 4 * c4 < N) {
                const float* src = B + (long)(k0 + kk) * ldb + 4 * c4;
                if (vecb) v = *reinterpret_cast<const f32x4*>(src);
                else {                         // ragged / unaligned rows of B (N not a multiple of 4): guarded scalars
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (4 * c4 + e < N) ? src[e] : 0.f;
                }
            }
            *reinterpret_cast<f32x4*>(Bs + kk * SB + 4 * c4) = v;
        }
        __syncthreads();
        mma_steps<NT, 1>(As, SA, 16 * wv, Bs, SB, m, g, RBK / 4, acc);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * wv + 4 * g + r, col = 16 * j + m;
            if (row < M && col < N) {
                float v = acc[0][j][r];
                if (relu) v = fmaxf(v, 0.f);
                C[(long)row * ldc + col] = v;
            }
        }
}

// ---- NN with the whole B resident (K <= 128: the ACM projections of a 128-feature input): a persistent workgroup per CU
// stages W once (106 KB at N = 192), then walks 64-row panels of A -- the next panel's 32 floats per thread are requested
// before the current panel feeds the matrix pipe, masked and parked in LDS after it.
template <int NT>
__global__ __launch_bounds__(256) void gemm_rows_nn_wres_kernel(int M, int N, int K, const float* __restrict__ A, long lda,
                                                                const float* __restrict__ B, long ldb, float* __restrict__ C, long ldc,
                                                                int relu, acm_dropout_t drop, int vecb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SA = RBM + 17;               // As[k][row]: 81 -- the staging writes (16 columns x 4 rows per wave) spread over the
                                               // banks (17 c + r), the operand reads of the two row groups stay apart
    const int SB = pad16(NT * 16);
    const int KP = (K + 63) / 64 * 64;         // K padded to whole 64-column groups (zero rows / columns)
    float* Ws = lds;                           // KP x SB
    float* As = lds + KP * SB;                 // KP x SA
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    const AcmDropCtx dc = acm_drop_ctx(drop);
    for (int idx = threadIdx.x; idx < KP * NT * 4; idx += 256) {
        const int kk = idx / (NT * 4), c4 = idx % (NT * 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kk < K && 4 * c4 < N) {
            const float* src = B + (long)kk * ldb + 4 * c4;
            if (vecb) v = *reinterpret_cast<const f32x4*>(src);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (4 * c4 + e < N) ? src[e] : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(Ws + kk * SB + 4 * c4) = v;
    }
    const int sc = threadIdx.x & 15, sr = threadIdx.x >> 4;
    const int npanels = (M + RBM - 1) / RBM;
    float av[2][RBM / 16][4];                  // [64-column group][row quarter][q]
    auto fetch = [&](int panel) {
#pragma unroll
        for (int grp = 0; grp < 2; ++grp)
#pragma unroll
            for (int i = 0; i < RBM / 16; ++i) {
                const int row = panel * RBM + sr + 16 * i;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = 64 * grp + sc + 16 * q;
                    av[grp][i][q] = (row < M && col < K) ? A[(long)row * lda + col] : 0.f;
                }
            }
    };
    auto park = [&](int panel) {
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            if (64 * grp >= KP) continue;
#pragma unroll
            for (int i = 0; i < RBM / 16; ++i) {
                const int r = sr + 16 * i, row = panel * RBM + r;
                if (dc.on) {
                    unsigned w[4];
                    acm_philox7(dc, row, sc + 16 * grp, w);
#pragma unroll
                    for (int q = 0; q < 4; ++q) av[grp][i][q] = (w[q] >= dc.thresh) ? av[grp][i][q] * dc.inv_keep : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) As[(64 * grp + sc + 16 * q) * SA + r] = av[grp][i][q];
            }
        }
    };
    int panel = blockIdx.x;
    if (panel < npanels) fetch(panel);
    for (; panel < npanels; panel += gridDim.x) {
        __syncthreads();                        // W staged (first round) / every wave done with the previous panel
        park(panel);
        __syncthreads();
        if (panel + (int)gridDim.x < npanels) fetch(panel + gridDim.x);
        f32x4 acc[1][NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_steps<NT, 1>(As, SA, 16 * wv, Ws, SB, m, g, KP / 4, acc);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = panel * RBM + 16 * wv + 4 * g + r, col = 16 * j + m;
                if (row < M && col < N) {
                    float v = acc[0][j][r];
                    if (relu) v = fmaxf(v, 0.f);
                    C[(long)row * ldc + col] = v;
                }
            }
    }
}

// ---- TN: slab[b][K, N] = drop(X)[rows of block b]^T dZ[rows of block b];  KT = ceil(K / 16) <= 8 row tiles of the output,
// wave w owns output rows 32 w .. 32 w + 31 (two row tiles) x all NT column tiles.
template <int NT>
__global__ __launch_bounds__(256) void gemm_rows_tn_kernel(int n_rows, int K, int N, const float* __restrict__ X, long ldx,
                                                           const float* __restrict__ Dz, long lddz, float* __restrict__ slabs,
                                                           int rows_per_block, acm_dropout_t drop, int vecb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int SX = 128 + 16;                   // Xs[row][f]: the contraction index (row) outermost
    const int SB = pad16(NT * 16);
    float* Xs = lds;                           // RBK x SX
    float* Bs = lds + RBK * SX;                // RBK x SB
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    const int r_begin = blockIdx.x * rows_per_block, r_end = min(n_rows, r_begin + rows_per_block);
    const AcmDropCtx dc = acm_drop_ctx(drop);
    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int sc = threadIdx.x & 15, sr = threadIdx.x >> 4;
    const bool wave_live = 32 * wv < K;        // output rows of this wave exist
    // One workgroup per CU (90 KB of LDS): the NEXT slab's global loads are issued into registers before the current slab
    // feeds the matrix pipe, masked and parked in LDS after it.
    float xv[RBK / 16][2][4];
    f32x4 bv[(RBK * NT * 4 + 255) / 256];
    auto fetch = [&](int r0) {
#pragma unroll
        for (int i = 0; i < RBK / 16; ++i) {
            const int row = r0 + sr + 16 * i;
#pragma unroll
            for (int grp = 0; grp < 2; ++grp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = 64 * grp + sc + 16 * q;
                    xv[i][grp][q] = (row < r_end && col < K) ? X[(long)row * ldx + col] : 0.f;
                }
        }
#pragma unroll
        for (int u = 0; u < (RBK * NT * 4 + 255) / 256; ++u) {
            const int idx = threadIdx.x + 256 * u, kk = idx / (NT * 4), c4 = idx % (NT * 4);
            bv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (idx < RBK * NT * 4 && r0 + kk < r_end && 4 * c4 < N) {
                const float* src = Dz + (long)(r0 + kk) * lddz + 4 * c4;
                if (vecb) bv[u] = *reinterpret_cast<const f32x4*>(src);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[u][e] = (4 * c4 + e < N) ? src[e] : 0.f;
                }
            }
        }
    };
    auto park = [&](int r0) {
#pragma unroll
        for (int i = 0; i < RBK / 16; ++i) {
            const int r = sr + 16 * i, row = r0 + r;
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                if (dc.on && 64 * grp < K) {
                    unsigned w[4];
                    acm_philox7(dc, row, sc + 16 * grp, w);
#pragma unroll
                    for (int q = 0; q < 4; ++q) xv[i][grp][q] = (w[q] >= dc.thresh) ? xv[i][grp][q] * dc.inv_keep : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) Xs[r * SX + 64 * grp + sc + 16 * q] = xv[i][grp][q];
            }
        }
#pragma unroll
        for (int u = 0; u < (RBK * NT * 4 + 255) / 256; ++u) {
            const int idx = threadIdx.x + 256 * u, kk = idx / (NT * 4), c4 = idx % (NT * 4);
            if (idx < RBK * NT * 4) *reinterpret_cast<f32x4*>(Bs + kk * SB + 4 * c4) = bv[u];
        }
    };
    if (r_begin < r_end) {
        fetch(r_begin);
        park(r_begin);
    }
    for (int r0 = r_begin; r0 < r_end; r0 += RBK) {
        __syncthreads();                        // the slab is in LDS
        const bool more = r0 + RBK < r_end;
        if (more) fetch(r0 + RBK);
        if (wave_live) mma_steps<NT, 2>(Xs, SX, 32 * wv, Bs, SB, m, g, RBK / 4, acc);
        __syncthreads();                        // every wave is done reading the slab
        if (more) park(r0 + RBK);
    }
    float* dst = slabs + (long)blockIdx.x * K * N;
    if (wave_live) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 32 * wv + 16 * i + 4 * g + r, col = 16 * j + m;
                    if (f < K && col < N) dst[(long)f * N + col] = acc[i][j][r];
                }
    }
}

}  // namespace

// Shapes these kernels are written for: up to 192 columns; rows of B / dZ that are 16-byte aligned multiples of 4 floats are
// staged with 16-byte loads, anything else with guarded scalars.
static bool rows_shape_ok(int64_t N) { return N >= 1 && N <= 192; }
static int rows_vecb(int64_t N, const float* B, int64_t ldb) { return N % 4 == 0 && ((uintptr_t)B) % 16 == 0 && ldb % 4 == 0; }

bool acm_gemm_rows_nn_ok(int64_t M, int64_t N, int64_t K, const float* B, int64_t ldb) {
    (void)B, (void)ldb;
    return M >= 4096 && K >= 16 && K <= 4096 && rows_shape_ok(N) && (acm_tuning().gemm_forms & (ACM_GEMM_ROWS | ACM_GEMM_ROWS_ALWAYS)) != 0;
}

int acm_gemm_rows_nn(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                     int64_t ldc, int relu, const acm_dropout_t* drop_in, hipStream_t st) {
    acm_dropout_t drop = {0.f, 0, 0, nullptr, 0, 0};
    if (drop_in) drop = *drop_in;
    const int nt = (int)((N + 15) / 16);
    int grid = (int)((M + RBM - 1) / RBM);
    const int ntr = nt <= 4 ? nt : (nt <= 6 ? 6 : (nt <= 8 ? 8 : (nt <= 10 ? 10 : 12)));      // the instantiated tile counts
    if (K <= 128) {          // B resident, persistent workgroups
        const int kp = (int)((K + 63) / 64 * 64);
        const size_t lds_w = ((size_t)kp * (((ntr * 16 + 31) / 32) * 32 + 16) + (size_t)kp * (RBM + 17)) * sizeof(float);
        const int blocks_per_cu = lds_w <= 48 * 1024 ? 3 : (lds_w <= 78 * 1024 ? 2 : 1);
        if (grid > 256 * blocks_per_cu) grid = 256 * blocks_per_cu;
#define ACM_RNW(NTv)                                                                                                    \
    do {                                                                                                                \
        ACM_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_rows_nn_wres_kernel<NTv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w)); \
        hipLaunchKernelGGL((gemm_rows_nn_wres_kernel<NTv>), dim3(grid), dim3(256), lds_w, st, (int)M, (int)N, (int)K, A, (long)lda, B, \
                           (long)ldb, C, (long)ldc, relu, drop, rows_vecb(N, B, ldb));                                  \
    } while (0)
        switch (ntr) {
            case 1: ACM_RNW(1); break;
            case 2: ACM_RNW(2); break;
            case 3: ACM_RNW(3); break;
            case 4: ACM_RNW(4); break;
            case 6: ACM_RNW(6); break;
            case 8: ACM_RNW(8); break;
            case 10: ACM_RNW(10); break;
            default: ACM_RNW(12); break;
        }
#undef ACM_RNW
        ACM_CHECK_HIP(hipGetLastError());
        return ACM_OK;
    }
    const size_t lds = ((size_t)RBK * (RBM + 16) + (size_t)RBK * (((ntr * 16 + 31) / 32) * 32 + 16)) * sizeof(float);
#define ACM_RNN(NTv)                                                                                                    \
    do {                                                                                                                \
        ACM_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_rows_nn_kernel<NTv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((gemm_rows_nn_kernel<NTv>), dim3(grid), dim3(256), lds, st, (int)M, (int)N, (int)K, A, (long)lda, B,  \
                           (long)ldb, C, (long)ldc, relu, drop, rows_vecb(N, B, ldb));                                                        \
    } while (0)
    switch (nt) {
        case 1: ACM_RNN(1); break;
        case 2: ACM_RNN(2); break;
        case 3: ACM_RNN(3); break;
        case 4: ACM_RNN(4); break;
        case 5: case 6: ACM_RNN(6); break;
        case 7: case 8: ACM_RNN(8); break;
        case 9: case 10: ACM_RNN(10); break;
        default: ACM_RNN(12); break;
    }
#undef ACM_RNN
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}

bool acm_gemm_rows_tn_ok(int64_t n_rows, int64_t K, int64_t N) {
    return n_rows >= 8192 && K >= 16 && K <= 128 && rows_shape_ok(N) && (acm_tuning().gemm_forms & (ACM_GEMM_ROWS | ACM_GEMM_ROWS_ALWAYS)) != 0;
}
int acm_gemm_rows_tn_blocks(int64_t n_rows) {
    int64_t nb = (n_rows + 4 * RBK - 1) / (4 * RBK);          // at least four slabs per workgroup
    return (int)(nb > 256 ? 256 : (nb < 1 ? 1 : nb));
}

// slabs: blocks x K x N floats; the caller reduces them (splitk_reduce_kernel of acm_gemm.hip)
int acm_gemm_rows_tn(int64_t n_rows, int64_t K, int64_t N, const float* X, int64_t ldx, const float* Dz, int64_t lddz,
                     float* slabs, int blocks, const acm_dropout_t* drop_in, hipStream_t st) {
    acm_dropout_t drop = {0.f, 0, 0, nullptr, 0, 0};
    if (drop_in) drop = *drop_in;
    const int nt = (int)((N + 15) / 16);
    int64_t rpb = (n_rows + blocks - 1) / blocks;
    rpb = (rpb + RBK - 1) / RBK * RBK;
    const size_t lds = ((size_t)RBK * (128 + 16) + (size_t)RBK * (((nt * 16 + 31) / 32) * 32 + 16)) * sizeof(float);
#define ACM_RTN(NTv)                                                                                                    \
    do {                                                                                                                \
        ACM_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_rows_tn_kernel<NTv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((gemm_rows_tn_kernel<NTv>), dim3(blocks), dim3(256), lds, st, (int)n_rows, (int)K, (int)N, X,   \
                           (long)ldx, Dz, (long)lddz, slabs, (int)rpb, drop, rows_vecb(N, Dz, lddz));                                           \
    } while (0)
    switch (nt) {
        case 1: ACM_RTN(1); break;
        case 2: ACM_RTN(2); break;
        case 3: ACM_RTN(3); break;
        case 4: ACM_RTN(4); break;
        case 5: case 6: ACM_RTN(6); break;
        case 7: case 8: ACM_RTN(8); break;
        case 9: case 10: ACM_RTN(10); break;
        default: ACM_RTN(12); break;
    }
#undef ACM_RTN
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}
