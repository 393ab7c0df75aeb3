// fp32 projections of a 128-feature input on the bf16 matrix pipe of gfx950, at fp32 accuracy (ACM-Geometric/layers.py:86-88,
// 101-103 and their MmBackward; the input dropout of models.py:54 drawn in the operand load).
//
// v_mfma_f32_16x16x4_f32 delivers 157 TFLOP/s on this part; the two projections of the arXiv-year-shaped graph
// (169 343 x 128 by 128 x 192: 8.3 GFLOP each) therefore cannot take less than 53 us on it, while their operands stream in
// ~45 us.  v_mfma_f32_16x16x32_bf16 is sixteen times as fast, and an fp32 number is EXACTLY the sum of three bf16 numbers
// (8 + 8 + 8 significand bits: x = hi + mid + lo with hi = trunc_bf16(x), mid = trunc_bf16(x - hi), lo = x - hi - mid, every
// subtraction exact).  A product x w then is the sum of nine bf16 products; the six with weight >= 2^-16 of the leading one
//     hi hi,  hi mid,  mid hi,  mid mid,  hi lo,  lo hi
// are kept (each exact in the fp32 accumulator), the three dropped ones (mid lo, lo mid, lo lo) are below 2^-24 of it -- under
// the rounding of an fp32 FMA chain.  Six bf16 MFMAs do the work of sixteen fp32 ones: the products become stream-bound.
//
//   NN   C  = drop(X) W       X: [n, K <= 128]   W: [K, N <= 192]
//        W sits in LDS for the whole launch, already split and laid out as MFMA operands (3 x 48 KB at N = 192); a wave takes
//        32 rows of X straight from global memory into operand registers (no LDS round trip, no barrier in the loop), draws
//        the mask, splits, and runs 6 MFMAs per (16 rows, 16 columns, 32 k).  The roles are transposed -- A operand = W^T,
//        B operand = X^T -- so that a lane ends up with FOUR CONSECUTIVE columns of one row of C: 16-byte stores.
//   TN   dW = drop(X)^T dZ    reduce over the n rows: X and dZ tiles are split once while they are staged, stored to LDS as
//        packed pairs of consecutive rows (the contraction index), so that an operand is one ds_read_b128.
//
// The contraction order inside a 32-k step is free as long as both operands use the same one; it is chosen so that the four
// columns c, c + 16, c + 32, c + 48 of a 64-column group -- the four words of ONE Philox call (acm_dropout.hip) -- sit in one
// lane: lane (g, row m) loads the 16-byte pieces at columns 64 G + 16 q + 4 g (G = 0..1, q = 0..3).
#include "acm_common.h"
#include "acm_bx3_device.h"

namespace {

// Where the columns of the product come from and go to.  b1 != nullptr: B = [B0 0 | B1 0 | B2], three [K, f] matrices read in
// place, the first two at a column pitch of fb >= f (zero columns between: acm_proj3).  split > 0: columns [0, split) to C,
// the rest to C2.
struct Bx3Cols {
    const float* b1;
    const float* b2;
    int f, fb, split;
    float* c2;
    long ldc2;
    int vecc2;
};

// ---- NN.  NT column tiles of 16 (even); 512 threads = 8 waves, each on its own panels of 16 NA rows (grid-stride).  The rows of
// the NEXT panel are requested before the current one feeds the matrix pipe (two register sets).
template <int NT, int NA>
__global__ __launch_bounds__(512, NT <= 4 ? 4 : 2) void gemm_bx3_nn_kernel(int M, int N, int K, const float* __restrict__ A, long lda,
                                                          const float* __restrict__ B, long ldb, float* __restrict__ C, long ldc,
                                                          int relu, acm_dropout_t drop, int vecc, Bx3Cols cs) {
    extern __shared__ __attribute__((aligned(16))) u32x4 Ws[];       // [part 3][tile NT][kb 4][lane 64]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    // W^T as the A operand: lane (g, i = m) of tile j, k block kb holds W[bx3_k(kb, g, e)][16 j + i], e = 0..7
    constexpr int WIT = (NT * 4 * 64 + 511) / 512;
    float wst[WIT][8];
#pragma unroll
    for (int it = 0; it < WIT; ++it) {         // every load of the block in flight before the first split
        const int idx = threadIdx.x + 512 * it;
        const int ln = idx & 63, kb = (idx >> 6) & 3, j = idx >> 8, gi = ln >> 4, col = 16 * j + (ln & 15);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kr = bx3_k(kb, gi, e);
            const float* bp = B;
            int cc = col;
            bool ok = idx < NT * 256 && kr < K && col < N;
            if (cs.b1) {                           // three matrices in place, channel blocks of fb columns
                const int blk = col < 2 * cs.fb ? col / cs.fb : 2;
                cc = col - blk * cs.fb;
                bp = blk == 0 ? B : (blk == 1 ? cs.b1 : cs.b2);
                ok = ok && cc < cs.f;
            }
            wst[it][e] = ok ? bp[(long)kr * ldb + cc] : 0.f;
        }
    }
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int idx = threadIdx.x + 512 * it;
        const int ln = idx & 63, kb = (idx >> 6) & 3, j = idx >> 8;
        u32x4 h, mdl, l;
        split3(wst[it], h, mdl, l);
        if (idx < NT * 256) {
            Ws[(0 * NT + j) * 256 + kb * 64 + ln] = h;
            Ws[(1 * NT + j) * 256 + kb * 64 + ln] = mdl;
            Ws[(2 * NT + j) * 256 + kb * 64 + ln] = l;
        }
    }
    const AcmDropCtx dc = acm_drop_ctx(drop);
    const int npan = (M + 16 * NA - 1) / (16 * NA), stride = gridDim.x * 8;
    f32x4 nxt[NA][2][4];
    auto fetch = [&](int pan) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            int row = pan * 16 * NA + 16 * p + m;
            row = row < M ? row : M - 1;                              // (results of such rows are not stored)
            const float* src = A + (long)row * lda + 4 * g;
#pragma unroll
            for (int G = 0; G < 2; ++G)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    nxt[p][G][q] = (64 * G + 16 * q + 4 * g < K) ? *reinterpret_cast<const f32x4*>(src + 64 * G + 16 * q)
                                                                               : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    int pan = blockIdx.x * 8 + wv;
    if (pan < npan) fetch(pan);
    __syncthreads();                               // W is in LDS
    for (; pan < npan; pan += stride) {
        f32x4 raw[NA][2][4];
#pragma unroll
        for (int p = 0; p < NA; ++p)
#pragma unroll
            for (int G = 0; G < 2; ++G)
#pragma unroll
                for (int q = 0; q < 4; ++q) raw[p][G][q] = nxt[p][G][q];
        fetch(pan + stride < npan ? pan + stride : pan);            // branch-free: the last round re-reads its own rows
        if (dc.on) {
#pragma unroll
            for (int p = 0; p < NA; ++p) {
                const int row = pan * 16 * NA + 16 * p + m;
#pragma unroll
                for (int G = 0; G < 2; ++G) {
                    if (64 * G >= K) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        unsigned w[4];
                        acm_philox7(dc, row, 4 * g + i + 16 * G, w);
#pragma unroll
                        for (int q = 0; q < 4; ++q) raw[p][G][q][i] = (w[q] >= dc.thresh) ? raw[p][G][q][i] * dc.inv_keep : 0.f;
                    }
                }
            }
        }
        f32x4 acc[NA][NT];
#pragma unroll
        for (int p = 0; p < NA; ++p)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[p][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (64 * (kb >> 1) >= K) continue;
            u32x4 xh[NA], xm[NA], xl[NA];
#pragma unroll
            for (int p = 0; p < NA; ++p) {
                const f32x4 u = raw[p][kb >> 1][2 * (kb & 1)], v = raw[p][kb >> 1][2 * (kb & 1) + 1];
                const float x8[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
                split3(x8, xh[p], xm[p], xl[p]);
            }
            // a step = two independent accumulators alternating (small terms first): the two row panels of one column tile
            // (NA = 2) or two column tiles of the one panel (NA = 1).  The W operands of the next step are read while this
            // one feeds the matrix pipe: two register sets, nothing further ahead.
            constexpr int TS = NA == 2 ? 1 : 2, STEPS = NT / TS;
            u32x4 wbuf[2][TS][3];
#pragma unroll
            for (int t = 0; t < TS; ++t)
#pragma unroll
                for (int part = 0; part < 3; ++part) wbuf[0][t][part] = Ws[(part * NT + t) * 256 + kb * 64 + lane];
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                if (s + 1 < STEPS) {
#pragma unroll
                    for (int t = 0; t < TS; ++t)
#pragma unroll
                        for (int part = 0; part < 3; ++part)
                            wbuf[(s + 1) & 1][t][part] = Ws[(part * NT + (s + 1) * TS + t) * 256 + kb * 64 + lane];
                }
                const int j0 = s * TS, j1 = NA == 2 ? j0 : j0 + 1, p1 = NA == 2 ? 1 : 0;
                const u32x4 ah = wbuf[s & 1][0][0], am = wbuf[s & 1][0][1], al = wbuf[s & 1][0][2];
                const u32x4 bh = wbuf[s & 1][TS - 1][0], bm = wbuf[s & 1][TS - 1][1], bl = wbuf[s & 1][TS - 1][2];
                // two accumulators alternate, small terms first (four chains -- the small terms in temporaries of their own --
                // measured slower: the step is bound by the LDS reads of W, not by the dependent issue)
                acc[0][j0] = mma(al, xh[0], acc[0][j0]);
                acc[p1][j1] = mma(bl, xh[p1], acc[p1][j1]);
                acc[0][j0] = mma(ah, xl[0], acc[0][j0]);
                acc[p1][j1] = mma(bh, xl[p1], acc[p1][j1]);
                acc[0][j0] = mma(am, xm[0], acc[0][j0]);
                acc[p1][j1] = mma(bm, xm[p1], acc[p1][j1]);
                acc[0][j0] = mma(am, xh[0], acc[0][j0]);
                acc[p1][j1] = mma(bm, xh[p1], acc[p1][j1]);
                acc[0][j0] = mma(ah, xm[0], acc[0][j0]);
                acc[p1][j1] = mma(bh, xm[p1], acc[p1][j1]);
                acc[0][j0] = mma(ah, xh[0], acc[0][j0]);
                acc[p1][j1] = mma(bh, xh[p1], acc[p1][j1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // D[i][jj]: lane (g, m), register r = C[row m of the panel][16 j + 4 g + r]
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const int row = pan * 16 * NA + 16 * p + m;
            if (row >= M) continue;
            float* dst = C + (long)row * ldc + 4 * g;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x4 v = acc[p][j];
                if (relu) v = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                const int col = 16 * j + 4 * g;
                if (cs.split > 0 && col + 3 >= cs.split) {          // (a group of four straddling the cut goes element-wise)
                    float* d2 = cs.c2 + (long)row * cs.ldc2;
                    if (col >= cs.split && cs.vecc2 && col + 3 < N) *reinterpret_cast<f32x4*>(d2 + (col - cs.split)) = v;
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (col + r >= N) continue;
                            if (col + r < cs.split) dst[16 * j + r] = v[r];
                            else d2[col + r - cs.split] = v[r];
                        }
                    }
                } else if (vecc && col + 3 < N) *reinterpret_cast<f32x4*>(dst + 16 * j) = v;
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < N) dst[16 * j + r] = v[r];
                }
            }
        }
    }
}

// ---- TN: slab[b][K, N] = drop(X)[rows of block b]^T dZ[rows of block b].  A 32-row slab of X (<= 128 columns) and dZ (16 NT
// columns) is split while it is staged and parked in LDS as T[part][column][row pairs] (16 dwords + 4 of padding per column:
// the operand reads -- lane (g, m): column 16 t + m, rows 8 g .. 8 g + 7 = one ds_read_b128 -- and the staging writes both
// fall on 64 distinct banks).  512 threads = TWO groups of four waves, each a pipeline of its own over every other slab with
// its own LDS buffer and a full set of accumulators (wave w of a group: output rows 32 w .. + 31, all columns): while one
// group feeds the matrix pipe from its buffer, the other splits and parks its next slab, so every SIMD holds one wave of
// each kind and the vector and matrix pipes overlap (with all eight waves in one phase a slab cost its VALU work PLUS its
// matrix work: 75 us for the 169 343 x 128 x 192 product).  The rows of a group's next slab are requested before it starts
// feeding the pipe.  The two groups' accumulators meet in LDS at the end.
constexpr int TS = 20;                     // dwords per column in LDS
// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the global loads in flight (s_waitcnt vmcnt(0)),
// i.e. the next slab's rows requested a phase ahead -- every phase would last a full memory latency.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int NT>
__global__ __launch_bounds__(512, 2) void gemm_bx3_tn_kernel(int n_rows, int Ktot, int N, const float* __restrict__ Xall, long ldx,
                                                             const float* __restrict__ Dz, long lddz, float* __restrict__ slabs,
                                                             int rows_per_block, acm_dropout_t drop) {
    extern __shared__ __attribute__((aligned(16))) unsigned Tl[];    // [group 2][part 3][128 + 16 NT columns][TS]
    constexpr int COLS = 128 + 16 * NT, BUF = 3 * COLS * TS, ZT = (64 * NT + 255) / 256;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    const int grp = wv >> 2, wg = wv & 3, tid = threadIdx.x & 255;
    // blockIdx.y: which 128 columns of X (= rows of the output) this workgroup owns
    const int k0 = 128 * blockIdx.y, K = min(128, Ktot - k0);
    const float* __restrict__ X = Xall + k0;
    const int r_begin = blockIdx.x * rows_per_block, r_end = min(n_rows, r_begin + rows_per_block);
    const int ns = (r_end - r_begin + 31) / 32, iters = (ns + 1) / 2;
    const AcmDropCtx dc = acm_drop_ctx(drop);
    unsigned* T = Tl + grp * BUF;
    // staging tasks of a group's 256 threads: X: (column 0..127) x (eight rows 8 rg .. 8 rg + 7), two per thread;
    // dZ: (column 0..16 NT - 1) x rg, ZT per thread
    float xv[2][8], zv[ZT][8];
    // per-task lane offsets (loop-invariant, 32-bit) and validity; the row part of an address is uniform: base + row * ld
    unsigned xoff[2], zoff[ZT];
    bool xok[2], zok[ZT];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = tid + 256 * u, xc = t & 127, xrg = t >> 7;
        xok[u] = xc < K;
        xoff[u] = (unsigned)(8 * xrg) * (unsigned)ldx + (unsigned)(xok[u] ? xc : 0);
    }
#pragma unroll
    for (int u = 0; u < ZT; ++u) {
        const int t = tid + 256 * u, zc = t % (16 * NT), zrg = t / (16 * NT);
        zok[u] = zrg < 4 && zc < N;
        zoff[u] = (unsigned)(8 * (zok[u] ? zrg : 0)) * (unsigned)lddz + (unsigned)(zok[u] ? zc : 0);
    }
    auto fetch = [&](int r0) {
        if (r0 + 32 <= r_end) {                    // whole slab: no row guards, uniform row bases + the lane's 32-bit offset
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float* xr = X + (long)(r0 + e) * ldx;
                const float* zr = Dz + (long)(r0 + e) * lddz;
#pragma unroll
                for (int u = 0; u < 2; ++u) xv[u][e] = xr[xoff[u]];
#pragma unroll
                for (int u = 0; u < ZT; ++u) zv[u][e] = zr[zoff[u]];
            }
            return;                                 // (columns beyond K / N are zeroed where the values are used: park)
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {              // the last, ragged slab of the matrix
            const int t = tid + 256 * u, xc = t & 127, xrg = t >> 7;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int row = r0 + 8 * xrg + e;
                xv[u][e] = (row < r_end && xc < K) ? X[(long)row * ldx + xc] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < ZT; ++u) {
            const int t = tid + 256 * u, zc = t % (16 * NT), zrg = t / (16 * NT);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int row = r0 + 8 * zrg + e;
                zv[u][e] = (zrg < 4 && row < r_end && zc < N) ? Dz[(long)row * lddz + zc] : 0.f;
            }
        }
    };
    auto park = [&](int r0) {
        u32x4 hi, md, lo;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tid + 256 * u, xc = t & 127, xrg = t >> 7;
            if (dc.on) {
                // lane l of a wave holds column 64 G + l: word l >> 4 of Philox(row, (l & 15) + 16 G).  Each lane draws TWO of
                // the eight rows (2 q, 2 q + 1 with q = l >> 4) and keeps 8 bits; four ds_bpermutes hand every lane its eight.
                const int q = lane >> 4, blockc = (lane & 15) + 16 * ((xc + k0) >> 6);
                unsigned bits = 0;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    unsigned w[4];
                    acm_philox7(dc, r0 + 8 * xrg + 2 * q + jj, blockc, w);
#pragma unroll
                    for (int v = 0; v < 4; ++v) bits |= (w[v] >= dc.thresh ? 1u : 0u) << (4 * jj + v);
                }
#pragma unroll
                for (int sq = 0; sq < 4; ++sq) {
                    const unsigned bb = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (16 * sq + (lane & 15)), (int)bits);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        xv[u][2 * sq + jj] = ((bb >> (4 * jj + q)) & 1u) ? xv[u][2 * sq + jj] * dc.inv_keep : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[u][e] = xok[u] ? xv[u][e] : 0.f;
            split3(xv[u], hi, md, lo);
            *reinterpret_cast<u32x4*>(T + (0 * COLS + xc) * TS + 4 * xrg) = hi;
            *reinterpret_cast<u32x4*>(T + (1 * COLS + xc) * TS + 4 * xrg) = md;
            *reinterpret_cast<u32x4*>(T + (2 * COLS + xc) * TS + 4 * xrg) = lo;
        }
#pragma unroll
        for (int u = 0; u < ZT; ++u) {
            const int t = tid + 256 * u, zc = t % (16 * NT), zrg = t / (16 * NT);
#pragma unroll
            for (int e = 0; e < 8; ++e) zv[u][e] = zok[u] ? zv[u][e] : 0.f;
            split3(zv[u], hi, md, lo);
            if (zrg < 4) {
                *reinterpret_cast<u32x4*>(T + (0 * COLS + 128 + zc) * TS + 4 * zrg) = hi;
                *reinterpret_cast<u32x4*>(T + (1 * COLS + 128 + zc) * TS + 4 * zrg) = md;
                *reinterpret_cast<u32x4*>(T + (2 * COLS + 128 + zc) * TS + 4 * zrg) = lo;
            }
        }
    };
    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int it0 = 2 * wg;
    const bool live = 16 * it0 < K;
    auto feed = [&]() {
        u32x4 a[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int part = 0; part < 3; ++part)
                a[i][part] = *reinterpret_cast<const u32x4*>(T + (part * COLS + 16 * (it0 + i) + m) * TS + 4 * g);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            u32x4 b[3];
#pragma unroll
            for (int part = 0; part < 3; ++part)
                b[part] = *reinterpret_cast<const u32x4*>(T + (part * COLS + 128 + 16 * j + m) * TS + 4 * g);
            // A operand = X^T (hi, mid, lo = a[i][0..2]), B operand = dZ; two accumulators alternate, small terms first
            acc[0][j] = mma(a[0][2], b[0], acc[0][j]);
            acc[1][j] = mma(a[1][2], b[0], acc[1][j]);
            acc[0][j] = mma(a[0][0], b[2], acc[0][j]);
            acc[1][j] = mma(a[1][0], b[2], acc[1][j]);
            acc[0][j] = mma(a[0][1], b[1], acc[0][j]);
            acc[1][j] = mma(a[1][1], b[1], acc[1][j]);
            acc[0][j] = mma(a[0][1], b[0], acc[0][j]);
            acc[1][j] = mma(a[1][1], b[0], acc[1][j]);
            acc[0][j] = mma(a[0][0], b[1], acc[0][j]);
            acc[1][j] = mma(a[1][0], b[1], acc[1][j]);
            acc[0][j] = mma(a[0][0], b[0], acc[0][j]);
            acc[1][j] = mma(a[1][0], b[0], acc[1][j]);
        }
    };
    // group 0 parks in even phases and feeds in odd ones, group 1 the other way round; every wave passes 2 iters + 1 barriers
    if (grp < ns) fetch(r_begin + 32 * grp);
    if (grp == 1) lds_barrier();
    for (int i = 0; i < iters; ++i) {
        const int sl = grp + 2 * i;
        if (sl < ns) park(r_begin + 32 * sl);
        lds_barrier();
        if (sl + 2 < ns) fetch(r_begin + 32 * (sl + 2));
        if (sl < ns && live) feed();
        lds_barrier();
    }
    if (grp == 0) lds_barrier();
    // the two groups' sums meet in LDS: group 1 writes [K rows][16 NT columns], group 0 adds and stores
    float* S = reinterpret_cast<float*>(Tl);
    if (grp == 1 && live) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) S[(16 * (it0 + i) + 4 * g + r) * (16 * NT) + 16 * j + m] = acc[i][j][r];
    }
    __syncthreads();
    float* dst = slabs + ((long)blockIdx.x * Ktot + k0) * N;
    if (grp == 0 && live) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 16 * (it0 + i) + 4 * g + r, col = 16 * j + m;
                    if (f < K && col < N) dst[(long)f * N + col] = acc[i][j][r] + S[f * (16 * NT) + col];
                }
    }
}

}  // namespace

// Shapes: tall X with at most 128 columns in 16-byte aligned rows (K a multiple of 4), at most 192 output columns.
bool acm_gemm_bx3_nn_ok(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda) {
    return M >= 8192 && K >= 32 && K <= 128 && K % 4 == 0 && N >= 1 && N <= 192 && lda % 4 == 0 && ((uintptr_t)A) % 16 == 0 &&
           (acm_tuning().gemm_forms & ACM_GEMM_BX3) != 0;
}

// w3 != nullptr: B, w3[0], w3[1] are the three [K, f] matrices of acm_proj3 (pitch ldb each), N = 2 fb + f
static int bx3_nn_ex(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                     int64_t ldc, int relu, const acm_dropout_t* drop_in, hipStream_t st, const float* const* w3, int f, int fb,
                     int64_t split, float* C2, int64_t ldc2) {
    Bx3Cols cs = {nullptr, nullptr, f, fb, (int)split, C2, (long)ldc2, 0};
    if (w3) cs.b1 = w3[0], cs.b2 = w3[1];
    cs.vecc2 = split > 0 && split % 4 == 0 && ldc2 % 4 == 0 && ((uintptr_t)C2) % 16 == 0;
    acm_dropout_t drop = {0.f, 0, 0, nullptr, 0, 0};
    if (drop_in) drop = *drop_in;
    const int nt = (int)((N + 15) / 16);
    const int ntr = nt <= 2 ? 2 : (nt <= 4 ? 4 : (nt <= 8 ? 8 : 12));
    const size_t lds = (size_t)3 * ntr * 4 * 64 * 16;
    const int per_cu = ntr <= 4 ? 2 : 1;                             // (<= 128 registers there: four waves per SIMD)
    const int na = 1;                                                // rows per wave step / 16
    const int64_t npan = (M + 16 * na - 1) / (16 * na);
    int grid = (int)((npan + 7) / 8);
    if (grid > 256 * per_cu) grid = 256 * per_cu;
    const int vecc = ldc % 4 == 0 && ((uintptr_t)C) % 16 == 0;
#define ACM_BX3(NTv, NAv)                                                                                                    \
    do {                                                                                                                \
        ACM_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bx3_nn_kernel<NTv, NAv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((gemm_bx3_nn_kernel<NTv, NAv>), dim3(grid), dim3(512), lds, st, (int)M, (int)N, (int)K, A, (long)lda, B,  \
                           (long)ldb, C, (long)ldc, relu, drop, vecc, cs);                                                \
    } while (0)
    switch (ntr) {
        case 2: ACM_BX3(2, 1); break;
        case 4: ACM_BX3(4, 1); break;
        case 8: ACM_BX3(8, 1); break;
        default: ACM_BX3(12, 1); break;
    }
#undef ACM_BX3
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}

// n_rows: the contraction length (rows of X and dZ); K: columns of X = rows of the output, in blocks of 128 per workgroup
bool acm_gemm_bx3_tn_ok(int64_t n_rows, int64_t K, int64_t N) {
    const int forms = acm_tuning().gemm_forms;
    if (!(forms & ACM_GEMM_BX3) || K < 32 || N < 1 || N > 192) return false;
    // wide inputs: from 16 384 rows (Penn94-like: 1 161 -> 625 us).  5 k rows (Squirrel) gain 7 us of 88, 2-3 k rows nothing.
    return K <= 128 ? n_rows >= 8192 : (n_rows >= 16384 && (forms & ACM_GEMM_BX3_WIDE) != 0);
}
// row ranges (one slab of the output per range): enough workgroups for the chip, at least four 32-row slabs each
int acm_gemm_bx3_tn_blocks(int64_t n_rows, int64_t K) {
    const int64_t kb = (K + 127) / 128;
    int64_t nb = (n_rows + 127) / 128, want = (512 + kb - 1) / kb;
    if (want > 256) want = 256;
    if (nb > want) nb = want;
    return (int)(nb < 1 ? 1 : nb);
}

// slabs: blocks x K x N floats; the caller reduces them (splitk_reduce_kernel of acm_gemm.hip)
int acm_gemm_bx3_tn(int64_t n_rows, int64_t K, int64_t N, const float* X, int64_t ldx, const float* Dz, int64_t lddz,
                    float* slabs, int blocks, const acm_dropout_t* drop_in, hipStream_t st) {
    acm_dropout_t drop = {0.f, 0, 0, nullptr, 0, 0};
    if (drop_in) drop = *drop_in;
    const int nt = (int)((N + 15) / 16);
    const int ntr = nt <= 2 ? 2 : (nt <= 4 ? 4 : (nt <= 8 ? 8 : 12));
    int64_t rpb = (n_rows + blocks - 1) / blocks;
    rpb = (rpb + 31) / 32 * 32;
    const size_t lds = (size_t)2 * 3 * (128 + 16 * ntr) * 20 * sizeof(unsigned);
#define ACM_BX3T(NTv)                                                                                                   \
    do {                                                                                                                \
        ACM_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bx3_tn_kernel<NTv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((gemm_bx3_tn_kernel<NTv>), dim3(blocks, (unsigned)((K + 127) / 128)), dim3(512), lds, st, (int)n_rows, (int)K, (int)N, X,  \
                           (long)ldx, Dz, (long)lddz, slabs, (int)rpb, drop);                                        \
    } while (0)
    switch (ntr) {
        case 2: ACM_BX3T(2); break;
        case 4: ACM_BX3T(4); break;
        case 8: ACM_BX3T(8); break;
        default: ACM_BX3T(12); break;
    }
#undef ACM_BX3T
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}

int acm_gemm_bx3_nn(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                    int64_t ldc, int relu, const acm_dropout_t* drop_in, hipStream_t st) {
    return bx3_nn_ex(M, N, K, A, lda, B, ldb, C, ldc, relu, drop_in, st, nullptr, 0, 0, 0, nullptr, 0);
}

// Z = relu?(drop?(X) [W_L 0 | W_H 0 | W_I]): header acm_proj3
extern "C" int acm_proj3(int64_t n_rows, int64_t K, const float* X, int64_t ldx, const float* w_low, const float* w_high,
                         const float* w_mlp, int64_t ld_w, int64_t f, int64_t f_block, float* C, int64_t ldc, int64_t split_col,
                         float* C2, int64_t ldc2, int relu, const acm_dropout_t* x_drop, acm_stream_t stream) {
    ACM_REQUIRE(X && w_low && w_high && w_mlp && C, ACM_EINVAL, "acm_proj3: NULL argument");
    ACM_REQUIRE(f >= 1 && f_block >= f && ld_w >= f && ldc >= 1 && split_col >= 0 && (split_col == 0 || (C2 && ldc2 >= 1)), ACM_ESHAPE,
                "acm_proj3: f %lld f_block %lld ld_w %lld split %lld", (long long)f, (long long)f_block, (long long)ld_w, (long long)split_col);
    const int64_t N = 2 * f_block + f;
    ACM_REQUIRE(split_col <= N, ACM_ESHAPE, "acm_proj3: split_col %lld > %lld columns", (long long)split_col, (long long)N);
    if (n_rows == 0) return ACM_OK;
    ACM_REQUIRE(acm_gemm_bx3_nn_ok(n_rows, N, K, X, ldx), ACM_EUNSUPPORTED,
                "acm_proj3: the split-bf16 row-panel kernel takes >= 8192 rows of 32..128 features (a multiple of 4, 16-byte aligned "
                "rows) and at most 192 product columns; got %lld x %lld -> %lld", (long long)n_rows, (long long)K, (long long)N);
    const float* w3[2] = {w_high, w_mlp};
    return bx3_nn_ex(n_rows, N, K, X, ldx, w_low, ld_w, C, ldc, relu, (x_drop && x_drop->p > 0.f) ? x_drop : nullptr,
                     (hipStream_t)stream, w3, (int)f, (int)f_block, split_col, C2, ldc2);
}

