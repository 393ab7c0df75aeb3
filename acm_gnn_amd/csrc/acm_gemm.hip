// K1 / K5: dense fp32 GEMM on the gfx950 f32 MFMA pipe (v_mfma_f32_16x16x4_f32: exact fp32,
// bitwise an fmaf chain in k order), for the per-channel projections of the ACM layer:
//     Z      = X * [W_L | W_H | W_I]          (N x F_in) * (F_in x 3F)      NN, optional ReLU
//     dWcat  = X^T * dZ                       (F_in x N) * (N x 3F)         TN, split-K over N
//     dX     = dZ * Wcat^T                    (N x 3F) * (3F x F_in)        NT
// Shapes are skinny and ragged (F_in = 7 .. 4814, 3F = 6 .. 192, N ~ 1e5), so every tile edge
// is guarded and there are three tile shapes:
//     64 x 64   (2x2 waves, 2x2 MFMA tiles each)   general
//     16 x 256  (1x4 waves, 1x4 tiles each)        M <= 16  (dW of a 7-feature input)
//     256 x 16  (4x1 waves, 4x1 tiles each)        N <= 16  (projection to 3*C classes)
// Tiles are staged through LDS (BK = 32) in whichever orientation keeps the *global* reads
// contiguous, with the next K-slab prefetched into registers while the current one feeds MFMA.
#include "acm_common.h"

// acm_gemm_rows.hip: row-panel kernels for n >> K, N (X read once, optional input dropout in the tile load)
bool acm_gemm_rows_nn_ok(int64_t M, int64_t N, int64_t K, const float* B, int64_t ldb);
int acm_gemm_rows_nn(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                     int64_t ldc, int relu, const acm_dropout_t* drop, hipStream_t st);
bool acm_gemm_rows_tn_ok(int64_t n_rows, int64_t K, int64_t N);
int acm_gemm_rows_tn_blocks(int64_t n_rows);
int acm_gemm_rows_tn(int64_t n_rows, int64_t K, int64_t N, const float* X, int64_t ldx, const float* Dz, int64_t lddz,
                     float* slabs, int blocks, const acm_dropout_t* drop, hipStream_t st);

// acm_gemm_bx3.hip: the same products on the bf16 matrix pipe at fp32 accuracy (three-way split operands), K <= 128
bool acm_gemm_bx3_nn_ok(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda);
int acm_gemm_bx3_nn(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                    int64_t ldc, int relu, const acm_dropout_t* drop, hipStream_t st);
bool acm_gemm_bx3_tn_ok(int64_t n_rows, int64_t K, int64_t N);
int acm_gemm_bx3_tn_blocks(int64_t n_rows, int64_t K);
int acm_gemm_bx3_tn(int64_t n_rows, int64_t K, int64_t N, const float* X, int64_t ldx, const float* Dz, int64_t lddz,
                    float* slabs, int blocks, const acm_dropout_t* drop, hipStream_t st);

int acm_linear_fwd_narrow(int64_t n_rows, int64_t f_in, int64_t f_out, const float* X, int64_t ldx, const float* W, int64_t ldw,
                          const float* bias, int relu, const acm_dropout_t* drop, float* Y, int64_t ldy, hipStream_t s,
                          const float* add, int64_t ld_add);   // acm_linear.hip

namespace {

constexpr int BK = 32;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS tile addressing.  KMAJOR: global memory is contiguous along k (A not transposed /
// B transposed): tile[mn][k], row stride BK + 1.  Otherwise contiguous along m/n:
// tile[k][mn], row stride BMN + 16 (so k and k + 1 land on different bank halves).
template <bool KMAJOR, int BMN>
struct TileLds {
    static constexpr int STRIDE = KMAJOR ? (BK + 1) : (BMN + 16);
    static constexpr int SIZE = KMAJOR ? BMN * STRIDE : BK * STRIDE;
    static __device__ __forceinline__ int at(int mn, int k) {
        return KMAJOR ? mn * STRIDE + k : k * STRIDE + mn;
    }
};

// Global -> register staging of one BMN x BK tile.  `gmn`/`gk` strides are in elements.
template <bool KMAJOR, int BMN>
struct TileLoad {
    static constexpr int PER_THREAD = BMN * BK / 256;
    float r[PER_THREAD];
    __device__ __forceinline__ void load(const float* __restrict__ base, long s_mn, long s_k, int mn0,
                                         int mn_lim, int k0, int k_lim) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int idx = threadIdx.x + i * 256;
            const int mn = KMAJOR ? idx / BK : idx % BMN;
            const int kk = KMAJOR ? idx % BK : idx / BMN;
            const int gmn = mn0 + mn, gk = k0 + kk;
            r[i] = (gmn < mn_lim && gk < k_lim) ? base[(long)gmn * s_mn + (long)gk * s_k] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int idx = threadIdx.x + i * 256;
            const int mn = KMAJOR ? idx / BK : idx % BMN;
            const int kk = KMAJOR ? idx % BK : idx / BMN;
            lds[TileLds<KMAJOR, BMN>::at(mn, kk)] = r[i];
        }
    }
};

// C tile = op(A) op(B) over k in [k_begin, k_end) of split blockIdx.z.
template <int WM, int WN, int WAVES_M, int WAVES_N, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(int M, int N, int K, const float* __restrict__ A, long lda,
                                                   const float* __restrict__ B, long ldb,
                                                   float* __restrict__ C, long ldc, int relu, int k_per_split,
                                                   float* __restrict__ slabs, int cb, long cbs, int split,
                                                   float* __restrict__ C2, long ldc2, const float* __restrict__ bias,
                                                   acm_dropout_t drop) {
    constexpr int BM = 16 * WM * WAVES_M, BN = 16 * WN * WAVES_N;
    constexpr bool A_KMAJOR = !TA, B_KMAJOR = TB;
    using LA = TileLds<A_KMAJOR, BM>;
    using LB = TileLds<B_KMAJOR, BN>;
    __shared__ float lds[LA::SIZE + LB::SIZE];
    float* As = lds;
    float* Bs = lds + LA::SIZE;

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wm = wv / WAVES_N, wn = wv % WAVES_N;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = blockIdx.z * k_per_split;
    const int k_end = min(K, k_begin + k_per_split);
    // element strides of op(A)[m][k] and op(B)[k][n] in memory
    const long a_sm = TA ? 1 : lda, a_sk = TA ? lda : 1;
    const long b_sn = TB ? ldb : 1, b_sk = TB ? 1 : ldb;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    TileLoad<A_KMAJOR, BM> ta;
    TileLoad<B_KMAJOR, BN> tb;
    if (k_begin < k_end) {
        ta.load(A, a_sm, a_sk, m0, M, k_begin, k_end);
        tb.load(B, b_sn, b_sk, n0, N, k_begin, k_end);
    }
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        ta.store(As);
        tb.store(Bs);
        __syncthreads();
        if (k0 + BK < k_end) {  // prefetch the next slab while this one feeds the MFMAs
            ta.load(A, a_sm, a_sk, m0, M, k0 + BK, k_end);
            tb.load(B, b_sn, b_sk, n0, N, k0 + BK, k_end);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int kk = ks * 4 + (lane >> 4);
            float a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = As[LA::at((wm * WM + i) * 16 + (lane & 15), kk)];
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = Bs[LB::at((wn * WN + j) * 16 + (lane & 15), kk)];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D fragment: col = lane & 15, row = (lane >> 4) * 4 + reg
    float* dst = C;
    long ldd = ldc;
    if (slabs) {
        dst = slabs + (long)blockIdx.z * M * N;
        ldd = N;
    }
    const AcmDropCtx dc = acm_drop_ctx(drop);             // epilogue of acm_linear_fwd: bias -> ReLU -> dropout
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + (wm * WM + i) * 16 + (lane >> 4) * 4 + r;
                const int col = n0 + (wn * WN + j) * 16 + (lane & 15);
                if (row < M && col < N) {
                    float v = acc[i][j][r];
                    if (!slabs) {
                        if (bias) v += bias[col];
                        if (relu) v = fmaxf(v, 0.f);
                        if (dc.on) v *= acm_drop1(dc, row, col);
                    }
                    if (slabs) dst[(long)row * ldd + col] = v;
                    else if (cb) dst[(long)(col / cb) * cbs + (long)row * ldd + (col % cb)] = v;   // column-block output
                    else if (split && col >= split) C2[(long)row * ldc2 + (col - split)] = v;       // two-matrix output
                    else dst[(long)row * ldd + col] = v;
                }
            }
}

// C[q] = sum_z slabs[z][q].  16 outputs x 16 split-lanes per block: each thread adds every 16th
// slab (coalesced 64 B segments across q), then the 16 split-lanes combine in a fixed LDS tree.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int M, int N, int splits,
                                                            const float* __restrict__ slabs,
                                                            float* __restrict__ C, long ldc, int relu, int cb, long cbs,
                                                            int split, float* __restrict__ C2, long ldc2,
                                                            const float* __restrict__ bias, acm_dropout_t drop) {
    __shared__ float red[16][17];
    const long total = (long)M * N;
    const int tq = threadIdx.x & 15, tz = threadIdx.x >> 4;
    const long q = (long)blockIdx.x * 16 + tq;
    float s = 0.f;
    if (q < total) {
        // eight independent loads in flight per thread (the loop is latency-bound otherwise: 64 dependent trips at
        // splits = 1024); the order of the additions is fixed, so the result is reproducible
        int z = tz;
        for (; z + 7 * 16 < splits; z += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slabs[(long)(z + 16 * u) * total + q];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < splits; z += 16) s += slabs[(long)z * total + q];
    }
    red[tz][tq] = s;
    __syncthreads();
    if (tz == 0 && q < total) {
        float t = 0.f;
#pragma unroll
        for (int z = 0; z < 16; ++z) t += red[z][tq];
        const long m = q / N;
        const int n = (int)(q % N);
        if (bias) t += bias[n];
        if (relu) t = fmaxf(t, 0.f);
        if (drop.p > 0.f) t *= acm_drop1(acm_drop_ctx(drop), m, n);
        if (cb) C[(long)(n / cb) * cbs + m * ldc + (n % cb)] = t;
        else if (split && n >= split) C2[m * ldc2 + (n - split)] = t;
        else C[m * ldc + n] = t;
    }
}

struct GemmPlan {
    int shape;  // 0: 64x64, 1: 16x256, 2: 256x16
    int bm, bn, splits, k_per_split;
    dim3 grid;
};

GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K) {
    GemmPlan p;
    if (N <= 16 && M > 16) {
        p.shape = 2; p.bm = 256; p.bn = 16;
    } else if (M <= 16) {
        p.shape = 1; p.bm = 16; p.bn = 256;
    } else {
        p.shape = 0; p.bm = 64; p.bn = 64;
    }
    const int64_t tm = (M + p.bm - 1) / p.bm, tn = (N + p.bn - 1) / p.bn;
    const int64_t tiles = tm * tn;
    int64_t splits = 1;
    if (tiles < 256 && K >= 8 * BK) {  // too few tiles to fill 256 CUs: split K
        const int64_t want = ((M * N <= 4096 ? 1024 : 512) + tiles - 1) / tiles;
        const int64_t kmax = (K + 4 * BK - 1) / (4 * BK);  // keep >= 4 slabs per split
        splits = want < kmax ? want : kmax;
        // tiny outputs (dW of a narrow layer) are pure K-streaming: more, shorter splits keep every CU
        // loading; the slab reduce stays cheap because M*N is small
        const int64_t cap = (M * N <= 4096) ? 1024 : 256;
        if (splits > cap) splits = cap;
        if (splits < 1) splits = 1;
    }
    int64_t kps = (K + splits - 1) / splits;
    kps = (kps + BK - 1) / BK * BK;
    if (kps < BK) kps = BK;
    splits = K > 0 ? (K + kps - 1) / kps : 1;
    p.splits = (int)splits;
    p.k_per_split = (int)kps;
    p.grid = dim3((unsigned)tn, (unsigned)tm, (unsigned)splits);
    return p;
}

template <int WM, int WN, int WVM, int WVN>
void launch_shape(int ta, int tb, const GemmPlan& p, hipStream_t st, int M, int N, int K, const float* A,
                  long lda, const float* B, long ldb, float* C, long ldc, int relu, float* slabs, int cb, long cbs,
                  int split, float* C2, long ldc2, const float* bias, const acm_dropout_t& drop) {
#define ACM_GEMM_LAUNCH(TAv, TBv)                                                                        \
    hipLaunchKernelGGL((gemm_kernel<WM, WN, WVM, WVN, TAv, TBv>), p.grid, dim3(256), 0, st, M, N, K, A, \
                       lda, B, ldb, C, ldc, relu, p.k_per_split, slabs, cb, cbs, split, C2, ldc2, bias, drop)
    if (!ta && !tb) ACM_GEMM_LAUNCH(false, false);
    else if (ta && !tb) ACM_GEMM_LAUNCH(true, false);
    else if (!ta && tb) ACM_GEMM_LAUNCH(false, true);
    else ACM_GEMM_LAUNCH(true, true);
#undef ACM_GEMM_LAUNCH
}

}  // namespace

// C[M, N] = A^T B over a tall contraction K (A: [K, M], B: [K, N]): which row-panel forms take the shape, and the slabs each
// writes (0 = does not apply).  ONE predicate for acm_gemm_workspace_bytes and for the dispatch in gemm_core.
struct TnPanel {
    int bx3_blocks, rows_blocks;
};
static TnPanel tn_panel(int64_t M, int64_t N, int64_t K) {
    TnPanel t{0, 0};
    if (K <= 0) return t;
    if (acm_gemm_bx3_tn_ok(K, M, N)) t.bx3_blocks = acm_gemm_bx3_tn_blocks(K, M);
    if (acm_gemm_rows_tn_ok(K, M, N)) t.rows_blocks = acm_gemm_rows_tn_blocks(K);
    return t;
}

extern "C" int acm_gemm_workspace_bytes(int transA, int transB, int64_t M, int64_t N, int64_t K,
                                        size_t* bytes) {
    (void)transA;
    (void)transB;
    ACM_REQUIRE(bytes, ACM_EINVAL, "acm_gemm_workspace_bytes: NULL argument");
    ACM_REQUIRE(M >= 0 && N >= 0 && K >= 0, ACM_ESHAPE, "acm_gemm_workspace_bytes: negative size");
    const GemmPlan p = plan_gemm(M, N, K);
    size_t need = p.splits > 1 ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0;
    if (transA && !transB) {                                 // the row-panel forms (may be taken): one slab per workgroup
        const TnPanel tp = tn_panel(M, N, K);
        const int nb = tp.bx3_blocks > tp.rows_blocks ? tp.bx3_blocks : tp.rows_blocks;
        const size_t rows = (size_t)nb * (size_t)M * (size_t)N * sizeof(float);
        need = rows > need ? rows : need;
    }
    *bytes = need;
    return ACM_OK;
}

extern "C" int acm_gemm(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A,
                        int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int relu,
                        void* workspace, size_t workspace_bytes, acm_stream_t stream) {
    return acm_gemm_blocks(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, 0, 0, relu, workspace, workspace_bytes, stream);
}

static int gemm_core(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                     int64_t ldb, float* C, int64_t ldc, int64_t c_col_block, int64_t c_block_stride, int64_t split_col,
                     float* C2, int64_t ldc2, int relu, void* workspace, size_t workspace_bytes, acm_stream_t stream,
                     const float* bias = nullptr, const acm_dropout_t* drop = nullptr, const acm_dropout_t* a_drop = nullptr);

// op(A) with the counter-based dropout applied to the STORED matrix A while its tiles are staged (element [r][c] of A as it
// lies in memory: the node-feature matrix X, whichever side of the product it is on):
//     transA = 0:  C = drop(A) B        (Z = dropout(X) W,        ACM-Geometric/models.py:54 + layers.py:86-88)
//     transA = 1:  C = drop(A)^T B      (dW = dropout(X)^T dZ,    the MmBackward of the same)
// Only the row-panel kernels carry it (A with many more rows than columns, N <= 192, ...): ACM_EUNSUPPORTED otherwise, and
// the caller applies acm_dropout itself.  a_drop NULL or p = 0: acm_gemm_blocks.
extern "C" int acm_gemm_drop(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                             const float* B, int64_t ldb, float* C, int64_t ldc, int64_t c_col_block, int64_t c_block_stride,
                             int relu, const acm_dropout_t* a_drop, void* workspace, size_t workspace_bytes, acm_stream_t stream) {
    return gemm_core(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, c_col_block, c_block_stride, 0, nullptr, 0, relu,
                     workspace, workspace_bytes, stream, nullptr, nullptr, (a_drop && a_drop->p > 0.f) ? a_drop : nullptr);
}

// Y = dropout(relu?(X W^T + b)): the residual branch of ACM-GCN++ (ACM-Geometric/models.py:26-27,55-56:
// F.dropout(F.relu(self.mlpX(x))) with mlpX = one nn.Linear) as one GEMM with the bias, the ReLU and the counter-based
// dropout in its epilogue.  W is in nn.Linear's layout [f_out, f_in].
extern "C" int acm_linear_fwd(int64_t n_rows, int64_t f_in, int64_t f_out, const float* X, int64_t ldx, const float* W,
                              int64_t ldw, const float* bias, int relu, const acm_dropout_t* drop, float* Y, int64_t ldy,
                              void* workspace, size_t workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(X && W && Y, ACM_EINVAL, "acm_linear_fwd: NULL argument");
    {   // a narrow input (the raw features of the ACM-GCN++ residual: f_in <= 16) streams instead (acm_linear.hip)
        const int st = acm_linear_fwd_narrow(n_rows, f_in, f_out, X, ldx, W, ldw, bias, relu, drop, Y, ldy, (hipStream_t)stream, nullptr, 0);
        if (st != ACM_EUNSUPPORTED) return st;
    }
    return gemm_core(0, 1, n_rows, f_out, f_in, X, ldx, W, ldw, Y, ldy, 0, 0, 0, nullptr, 0, relu, workspace, workspace_bytes,
                     stream, bias, drop);
}

extern "C" int acm_gemm_blocks(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A,
                               int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t c_col_block,
                               int64_t c_block_stride, int relu, void* workspace, size_t workspace_bytes,
                               acm_stream_t stream) {
    return gemm_core(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, c_col_block, c_block_stride, 0, nullptr, 0, relu,
                     workspace, workspace_bytes, stream);
}

extern "C" int acm_gemm_split(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                              const float* B, int64_t ldb, float* C, int64_t ldc, int64_t split_col, float* C2,
                              int64_t ldc2, int relu, void* workspace, size_t workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(split_col > 0 && split_col < N && C2 && ldc2 >= N - split_col && ldc >= split_col, ACM_ESHAPE,
                "acm_gemm_split: split %lld of %lld columns, ldc %lld ldc2 %lld", (long long)split_col, (long long)N,
                (long long)ldc, (long long)ldc2);
    return gemm_core(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, 0, 0, split_col, C2, ldc2, relu, workspace,
                     workspace_bytes, stream);
}

static int gemm_core(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                     int64_t ldb, float* C, int64_t ldc, int64_t c_col_block, int64_t c_block_stride, int64_t split_col,
                     float* C2, int64_t ldc2, int relu, void* workspace, size_t workspace_bytes, acm_stream_t stream,
                     const float* bias, const acm_dropout_t* drop_in, const acm_dropout_t* a_drop) {
    acm_dropout_t drop = {0.f, 0, 0, nullptr, 0};
    if (drop_in) drop = *drop_in;
    ACM_REQUIRE(drop.p == 0.f || (drop.p > 0.f && drop.p < 1.f && drop.step), ACM_EINVAL, "acm_gemm: bad dropout spec");
    ACM_REQUIRE(M >= 0 && N >= 0 && K >= 0, ACM_ESHAPE, "acm_gemm: negative size");
    ACM_REQUIRE(c_col_block >= 0 && c_col_block < INT32_MAX, ACM_ESHAPE, "acm_gemm: bad column block");
    const int cb = (int)c_col_block;
    const long cbs = (long)c_block_stride;
    ACM_REQUIRE(M < INT32_MAX && N < INT32_MAX && K < INT32_MAX, ACM_EUNSUPPORTED, "acm_gemm: size >= 2^31");
    if (M == 0 || N == 0) return ACM_OK;
    ACM_REQUIRE(C && (K == 0 || (A && B)), ACM_EINVAL, "acm_gemm: NULL matrix pointer");
    const int split = (int)split_col;
    ACM_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) &&
                    ldc >= (cb ? (cb < N ? cb : N) : (split ? split : N)), ACM_ESHAPE,
                "acm_gemm: leading dimension too small (lda %lld ldb %lld ldc %lld)", (long long)lda,
                (long long)ldb, (long long)ldc);
    hipStream_t st = (hipStream_t)stream;
    ACM_REQUIRE(!a_drop || (a_drop->p > 0.f && a_drop->p < 1.f && a_drop->step), ACM_EINVAL, "acm_gemm_drop: bad dropout spec");
    // row-panel forms (acm_gemm_rows.hip): A = the tall node-feature matrix, read exactly once
    const bool plain_out = !split && !bias && drop.p == 0.f;
    // Measured on the arXiv-year projection shapes (scripts/probe_gemm_rows.py -> profiles/r03_gemm_rows.txt), us, row-panel
    // against tile kernel:  NN 169343 x 192 x 128: 140 / 131;  x 21: 31 / 44;  TN 128 x 192 x 169343: 127 / 150;  x 41554: 53 / 41
    // -- one wave per SIMD serialises matrix pipe (43 % busy), staging and stores in the row-panel NN.  So without a dropout
    // to carry, the row-panel forms take only the shapes they win: narrow outputs (NN), very tall contractions (TN).
    if (K > 0 && !transA && !transB && plain_out && !cb && acm_gemm_bx3_nn_ok(M, N, K, A, lda))
        return acm_gemm_bx3_nn(M, N, K, A, lda, B, ldb, C, ldc, relu, a_drop, st);
    if (K > 0 && !transA && !transB && plain_out && !cb && acm_gemm_rows_nn_ok(M, N, K, B, ldb) &&
        (a_drop || N <= 64 || (acm_tuning().gemm_forms & ACM_GEMM_ROWS_ALWAYS)))
        return acm_gemm_rows_nn(M, N, K, A, lda, B, ldb, C, ldc, relu, a_drop, st);
    const TnPanel tp = (transA && !transB && plain_out) ? tn_panel(M, N, K) : TnPanel{0, 0};
    const bool bx3_tn = tp.bx3_blocks > 0;
    if (bx3_tn || (tp.rows_blocks > 0 && (a_drop || K >= 100000 || (acm_tuning().gemm_forms & ACM_GEMM_ROWS_ALWAYS)))) {
        const int blocks = bx3_tn ? tp.bx3_blocks : tp.rows_blocks;
        const size_t need = (size_t)blocks * (size_t)M * (size_t)N * sizeof(float);
        ACM_REQUIRE(workspace && workspace_bytes >= need, ACM_ENOMEM, "acm_gemm: workspace %zu B < required %zu B", workspace_bytes, need);
        int rc = bx3_tn ? acm_gemm_bx3_tn(K, M, N, A, lda, B, ldb, (float*)workspace, blocks, a_drop, st)
                        : acm_gemm_rows_tn(K, M, N, A, lda, B, ldb, (float*)workspace, blocks, a_drop, st);
        if (rc != ACM_OK) return rc;
        const long total = (long)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, (int)M, (int)N, blocks,
                           (const float*)workspace, C, (long)ldc, relu, cb, cbs, 0, (float*)nullptr, 0L, (const float*)nullptr, drop);
        ACM_CHECK_HIP(hipGetLastError());
        return ACM_OK;
    }
    ACM_REQUIRE(!a_drop, ACM_EUNSUPPORTED, "acm_gemm_drop: the dropout in the tile load exists in the row-panel kernels only "
                "(tall A, N <= 192; at most 128 columns of A for the transposed product)");
    const GemmPlan p = plan_gemm(M, N, K);
    ACM_REQUIRE(p.grid.y <= 65535 && p.grid.z <= 65535, ACM_EUNSUPPORTED, "acm_gemm: grid too large");
    float* slabs = nullptr;
    if (p.splits > 1) {
        const size_t need = (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float);
        ACM_REQUIRE(workspace && workspace_bytes >= need, ACM_ENOMEM,
                    "acm_gemm: workspace %zu B < required %zu B", workspace_bytes, need);
        slabs = (float*)workspace;
    }
    ACM_REQUIRE(K > 0 || (!bias && drop.p == 0.f), ACM_EUNSUPPORTED, "acm_linear_fwd: f_in must be positive");
    if (K == 0) {  // empty sum
        if (split) {
            ACM_CHECK_HIP(hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)split * sizeof(float), (size_t)M, st));
            ACM_CHECK_HIP(hipMemset2DAsync(C2, (size_t)ldc2 * sizeof(float), 0, (size_t)(N - split) * sizeof(float), (size_t)M, st));
        } else if (!cb) {
            ACM_CHECK_HIP(hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)M, st));
        } else {
            for (int64_t n0 = 0; n0 < N; n0 += cb)
                ACM_CHECK_HIP(hipMemset2DAsync(C + (n0 / cb) * cbs, (size_t)ldc * sizeof(float), 0,
                                               (size_t)(N - n0 < cb ? N - n0 : cb) * sizeof(float), (size_t)M, st));
        }
        return ACM_OK;
    }
    if (p.shape == 0)
        launch_shape<2, 2, 2, 2>(transA, transB, p, st, (int)M, (int)N, (int)K, A, lda, B, ldb, C, ldc, relu, slabs, cb, cbs, split, C2, (long)ldc2, bias, drop);
    else if (p.shape == 1)
        launch_shape<1, 4, 1, 4>(transA, transB, p, st, (int)M, (int)N, (int)K, A, lda, B, ldb, C, ldc, relu, slabs, cb, cbs, split, C2, (long)ldc2, bias, drop);
    else
        launch_shape<4, 1, 4, 1>(transA, transB, p, st, (int)M, (int)N, (int)K, A, lda, B, ldb, C, ldc, relu, slabs, cb, cbs, split, C2, (long)ldc2, bias, drop);
    ACM_CHECK_HIP(hipGetLastError());
    if (slabs) {
        const long total = (long)M * N;
        const int grid = (int)((total + 15) / 16);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, (int)M, (int)N, p.splits, slabs,
                           C, (long)ldc, relu, cb, cbs, split, C2, (long)ldc2, bias, drop);
        ACM_CHECK_HIP(hipGetLastError());
    }
    return ACM_OK;
}
