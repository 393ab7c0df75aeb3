// The whole training step of the two-layer ACM model on a SMALL graph in six launches (acm_small_step, ABI 25).
//
// Reference: one step of ACM-Pytorch/models/models.py:100-166 (dropout -> GraphConvolution -> relu -> dropout ->
// GraphConvolution), utils.py:547-574 (log_softmax + nll_loss on the training rows, backward, optimizer.step) --
// layers.py:154-232 / ACM-Geometric/layers.py:78-116 per layer -- ~250 ATen launches there, 17-18 launches on this
// library's general path, whose kernels are sized for graphs 100x larger.  Cora (2 708 nodes), Chameleon (2 277),
// Squirrel (5 201): every table of the step fits one XCD's L2, a launch costs >= 5 us whatever it does, and these
// graphs are trained for thousands of epochs x 10 splits (ACM-Pytorch/train.py:95-139).  So: ONE launch per dependency
// level of the step (a level ends where the next one needs rows of OTHER nodes), everything row-local folded into the
// gather that produces its input, every update applied where its gradient becomes final.
//
// Layout conventions (wave = 64 lanes = four 16-lane groups g, lane m of a group):
//   * wide rows (64 columns): lane (g, m) owns columns 4m .. 4m+3 -- one 16-byte load per lane and row.  In a GATHER group
//     g takes every fourth neighbour (one wave instruction fetches four 256-byte rows); in ROW MATH group g owns channel g
//     (low, high, identity, structure): the per-channel LayerNorm / attention sums are 16-lane DPP sums, the 4 (or 3)
//     channels run side by side, per-row scalars travel through v_readlane.
//   * narrow rows (<= 8 columns per channel): tables of 16-byte blocks [L 8 | H 8 | S 8]; lane (e, q) fetches block q of
//     the neighbour in slot e (eight neighbours per wave instruction); row math runs with lane c = column c.
//   * long rows: the handle's work items (acm_csr.cpp: build_items) cut them into pieces; a piece leaves its partial sums
//     in its slot (write-through stores), arrives at the row's counter, and the LAST piece to arrive adds the slots in slot
//     order and runs the row's epilogue -- no second launch, no float atomics, deterministic.
#include <algorithm>

#include "acm_common.h"
#include "acm_adam_device.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int F = 64;                 // hidden width
constexpr int C8 = 8;                 // padded class count
constexpr int WAVES = 8;              // per workgroup of the gather phases (two per SIMD: the second hides the first's latency chain)
constexpr int MAX_WG = 512;           // workgroups of the phases that leave per-workgroup partial sums (launches 3, 4): beyond that a
                                      // wave loops over work items (Squirrel: 875 -> 512 workgroups, launch 4 56 -> 47 us)
constexpr int MAX_WG_WIDE = 2048;     // ... of the wide gather phases (launches 2, 5): one work item per wave up to 16384 items (a
                                      // wave's item is a chain of ~2 000 dependent instructions; more waves, not longer loops, hide it)
constexpr int RED_EL = 8;             // elements of the partial-sum vectors per reducing block of the last launch (x 32 producer lanes)
constexpr int PART4 = 3 * F * C8 + 3 * 4 * F + 16;      // dW2 [3][64][8] | dv1 [4][64] | dgamma1 | dbeta1 | dmix1 [4][4]
constexpr int PART3 = 3 * 4 * C8 + 16 + 1;              // dv2 [4][8] | dgamma2 | dbeta2 | dmix2 | loss
constexpr int PART3_PITCH = 128;
constexpr int LONG4 = PART4;                            // a long row's record of launch 4: the row's terms in the layout of a partial

struct SmallTensor {
    float* p;
    float* g;
    float* m;
    float* v;
    float* step;
};

struct ItemView {
    const AcmItem* items;
    int n_items;
    const AcmLongRow* long_rows;
    const int32_t* long_index;
    const int32_t* indices;
};

struct SmallDev {
    int n, f_in, C, k, relu_before, layernorm, train, update;
    float scale;
    SmallTensor t[2][ACM_SMALL_ROLES];
    ItemView graph, x, xt;
    const float* x_vals;
    const int32_t* xt_src_pos;
    const float* row_scale;
    const int64_t* labels;
    const float* row_weight;
    float* loss;
    float* logits;
    float *att1, *att2;
    // workspace
    float *Z1, *H1, *ST1, *OUT1, *T2, *Z2I, *G2, *DZ2, *G1, *DZ1, *slots, *part3, *part4, *W2T, *FACT;
    int64_t* latch;                  // the dropout counter as launch 1 found it: what launches 2-6 draw their masks with
    float *long3, *long4;            // [n_long][PART3_PITCH] / [n_long][LONG4]: the parameter-gradient terms of the long rows
    int n_long;
    const float* xt_vals;
    int* counters;
    int nwg3, nwg4;                  // producer workgroups of the two partial-sum buffers
    acm_dropout_t drop_in, drop_hidden;
    AdamScalars hp;
    int64_t* also_advance;
    int* arrive;
};

__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
__device__ __forceinline__ f4 zero4() { return f4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f4 relu4(f4 v) { return f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; }
__device__ __forceinline__ float hsum4(f4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ f4 sel4(int g, f4 a, f4 b, f4 c, f4 d) { return g == 0 ? a : (g == 1 ? b : (g == 2 ? c : d)); }
__device__ __forceinline__ f4 xsum4(f4 v) {      // over the four 16-lane groups
    return f4{acm_cross_row_sum(v.x), acm_cross_row_sum(v.y), acm_cross_row_sum(v.z), acm_cross_row_sum(v.w)};
}
__device__ __forceinline__ float bcast_f(float v, int u) { return __int_as_float(acm_row_bcast(__float_as_int(v), u)); }

// slot traffic crosses workgroups (and XCDs: each has its own L2): write-through stores, cache-bypassing loads
__device__ __forceinline__ void st_coherent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_coherent(const float* p) {
    return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st4_coherent(float* p, f4 v) {
    st_coherent(p, v.x), st_coherent(p + 1, v.y), st_coherent(p + 2, v.z), st_coherent(p + 3, v.w);
}
__device__ __forceinline__ f4 ld4_coherent(const float* p) {
    return f4{ld_coherent(p), ld_coherent(p + 1), ld_coherent(p + 2), ld_coherent(p + 3)};
}

// ------------------------------------------------------------------------------------------------ wide gather
// acc[c] += sum over the entries [begin, end) of an id list of  w(pos) * T_c[id(pos), 4m .. 4m+3];  group g of the wave
// takes the entries 4u + g.  The ids are loaded TRANSPOSED (lane (g, u) holds the id step u needs in group g), so a step
// is one row-broadcast DPP move per lane, no LDS crossbar.  Entries past `end` read row `safe` with weight 0.
template <int NCH, int RELU_MASK, class W>
__device__ __forceinline__ void wide_gather(const int32_t* __restrict__ ids, int begin, int end, const float* const (&tab)[3],
                                            const int (&ld)[3], int safe, W&& weight, f4 (&acc)[3]) {
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    int id = safe;
    float w = 0.f;
    if (begin + 4 * m + g < end) id = ids[begin + 4 * m + g], w = weight(begin + 4 * m + g);
    for (int k0 = begin; k0 < end; k0 += 64) {
        // the NEXT chunk's ids travel while this chunk's rows do (one dependent round trip less per 64 entries)
        const int npos = k0 + 64 + 4 * m + g;
        int nid = safe;
        float nw = 0.f;
        if (npos < end) nid = ids[npos], nw = weight(npos);
        const int steps = (min(end - k0, 64) + 3) >> 2;              // uniform
        // CNT steps (x NCH rows) requested before the first is consumed: a step is a dependent L2 round trip, and with one
        // item per wave nothing else hides it
#define ACM_WG_BLOCK(U0, CNT)                                                                   \
        {                                                                                       \
            f4 v[CNT][NCH];                                                                     \
            float ww[CNT];                                                                      \
            _Pragma("unroll") for (int s = 0; s < CNT; ++s) {                                   \
                const int j = acm_row_bcast(id, U0 + s);                                        \
                ww[s] = bcast_f(w, U0 + s);                                                     \
                _Pragma("unroll") for (int c = 0; c < NCH; ++c)                                 \
                    v[s][c] = ld4(tab[c] + (long)j * ld[c] + 4 * m);                            \
            }                                                                                   \
            _Pragma("unroll") for (int s = 0; s < CNT; ++s)                                     \
                _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                               \
                    f4 x = v[s][c];                                                             \
                    if ((RELU_MASK >> c) & 1) x = relu4(x);                                     \
                    acc[c] += ww[s] * x;                                                        \
                }                                                                               \
        }
        if (steps <= 4) {
            ACM_WG_BLOCK(0, 4)
        } else if (steps <= 8) {
            ACM_WG_BLOCK(0, 8)
        } else {
            ACM_WG_BLOCK(0, 8)
            if (steps <= 12) {
                ACM_WG_BLOCK(8, 4)
            } else {
                ACM_WG_BLOCK(8, 8)
            }
        }
#undef ACM_WG_BLOCK
        id = nid, w = nw;
    }
}

// The item loop of a wide phase: every wave takes the items wave, wave + n_waves, ...; `gather(item, acc)` fills the lane's
// partial sums, `epi(row, acc)` runs once per ROW with the complete sums (in every group).
// `pre(row)`: the loads of the row's epilogue that depend on nothing but the row number, requested BEFORE the gather (they
// travel while the item -> ids -> rows chain runs; with one item per wave that chain is the kernel's duration).
template <int NCH, class Pre, class Gather, class Epi>
__device__ __forceinline__ void wide_items(const ItemView& iv, float* slots, int* counters, int wave, int n_waves, Pre&& pre,
                                           Gather&& gather, Epi&& epi) {
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    for (int it = wave; it < iv.n_items; it += n_waves) {
        AcmItem item = iv.items[it];
        item.row = acm_uniform(item.row), item.begin = acm_uniform(item.begin), item.end = acm_uniform(item.end), item.slot = acm_uniform(item.slot);
        auto pf = pre(item.row);
        f4 acc[3] = {zero4(), zero4(), zero4()};
        gather(item, acc);
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = xsum4(acc[c]);
        if (item.slot >= 0) {                                        // a piece of a long row (uniform branch)
            float* sp = slots + (long)item.slot * (3 * F);
            if (g < NCH) st4_coherent(sp + g * F + 4 * m, sel4(g, acc[0], acc[1], acc[2], acc[2]));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int li = iv.long_index[item.row];
            const AcmLongRow lr = iv.long_rows[li];
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(counters + li, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != lr.slot_end - lr.slot_begin - 1) continue;    // another piece will finish the row
            if (lane == 0) __hip_atomic_store(counters + li, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc[0] = acc[1] = acc[2] = zero4();
            for (int q = lr.slot_begin; q < lr.slot_end; ++q) {
                const float* qp = slots + (long)q * (3 * F) + 4 * m;
#pragma unroll
                for (int c = 0; c < NCH; ++c) acc[c] += ld4_coherent(qp + c * F);
            }
        }
        epi(item.row, acc, pf);
    }
}

// ------------------------------------------------------------------------------------------------ narrow gather
// Rows of NQ 16-byte blocks (NQ = 4: [L 8 | H 8], 6: + [S 8]); lane = 8 e + q fetches block q of the neighbour in slot e.
// Returns in EVERY lane (e, q) block q of the row sum.  `relu01`: ReLU on the gathered L / H blocks (ACMII).
template <int NQ>
__device__ __forceinline__ f4 narrow_gather(const int32_t* __restrict__ ids, int begin, int end, const float* __restrict__ tab,
                                            int ld, int safe, bool relu01) {
    const int lane = threadIdx.x & 63, q = lane & 7, base = lane & ~7;
    const int qq = q < NQ ? q : NQ - 1;                  // idle lanes repeat the last block (unconditional loads: a guarded load
    const bool relu = relu01 && q < 4;                   // makes the compiler wait for every one before the next)
    f4 acc = zero4();
    int id = safe;
    float w = 0.f;
    if (begin + 8 * q + (lane >> 3) < end) id = ids[begin + 8 * q + (lane >> 3)], w = 1.f;
    for (int k0 = begin; k0 < end; k0 += 64) {
        const int npos = k0 + 64 + 8 * q + (lane >> 3);       // the next chunk's ids, requested before this chunk's rows
        int nid = safe;
        float nw = 0.f;
        if (npos < end) nid = ids[npos], nw = 1.f;
        const int steps = (min(end - k0, 64) + 7) >> 3;
        f4 v[8];
        float ww[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u < steps || u < 2) {                    // (uniform; two steps always: most rows of a small graph end there)
                const int j = __shfl(id, base + u);
                ww[u] = __shfl(w, base + u);
                v[u] = ld4(tab + (long)j * ld + 4 * qq);
            } else {
                v[u] = zero4(), ww[u] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f4 x = v[u];
            if (relu) x = relu4(x);
            acc += ww[u] * x;
        }
        id = nid, w = nw;
    }
    if (q >= NQ) acc = zero4();
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off);
        acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off);
        acc.w += __shfl_xor(acc.w, off);
    }
    return acc;
}

// The epilogue also receives the row's long-row index (-1: a whole row).  WHICH wave finishes a long row depends on the
// arrival order, so a finisher must not add the row's parameter-gradient terms to its own running sums (the sums of the
// workgroups would differ from run to run in the last bit): it leaves them in the row's own record (SmallDev::long3 / long4).
template <int NQ, class Pre, class Gather, class Epi>
__device__ __forceinline__ void narrow_items(const ItemView& iv, float* slots, int* counters, int wave, int n_waves, Pre&& pre,
                                             Gather&& gather, Epi&& epi) {
    const int lane = threadIdx.x & 63, q = lane & 7;
    for (int it = wave; it < iv.n_items; it += n_waves) {
        AcmItem item = iv.items[it];
        item.row = acm_uniform(item.row), item.begin = acm_uniform(item.begin), item.end = acm_uniform(item.end), item.slot = acm_uniform(item.slot);
        auto pf = pre(item.row);
        f4 acc = gather(item);
        int li = -1;
        if (item.slot >= 0) {
            float* sp = slots + (long)item.slot * (3 * F);
            if (lane < 8 && q < NQ) st4_coherent(sp + 4 * q, acc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            li = iv.long_index[item.row];
            const AcmLongRow lr = iv.long_rows[li];
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(counters + li, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != lr.slot_end - lr.slot_begin - 1) continue;
            if (lane == 0) __hip_atomic_store(counters + li, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc = zero4();
            if (q < NQ)
                for (int s = lr.slot_begin; s < lr.slot_end; ++s) acc += ld4_coherent(slots + (long)s * (3 * F) + 4 * q);
        }
        epi(item.row, acc, pf, li);
    }
}

// column c = lane & 7 of channel block `blk` (0: L, 1: H, 2: S) out of the per-lane 16-byte blocks of a narrow row
__device__ __forceinline__ float narrow_pick(f4 acc, int blk) {
    const int lane = threadIdx.x & 63, c = lane & 7, src = 2 * blk + (c >> 2);
    const float x0 = __shfl(acc.x, src), x1 = __shfl(acc.y, src), x2 = __shfl(acc.z, src), x3 = __shfl(acc.w, src);
    const int el = c & 3;
    return el == 0 ? x0 : (el == 1 ? x1 : (el == 2 ? x2 : x3));
}
__device__ __forceinline__ float gsum8(float v) { return acm_group_sum<8>(v); }
__device__ __forceinline__ float gmax8(float v) {
    v = fmaxf(v, __shfl_xor(v, 1));
    v = fmaxf(v, __shfl_xor(v, 2));
    return fmaxf(v, __shfl_xor(v, 4));
}

// ------------------------------------------------------------------------------------------------ Adam in an epilogue
// the update of four consecutive elements whose gradient `g` has just become final
__device__ __forceinline__ void adam_four(const SmallTensor& t, long i, f4 g, const AdamFactors& f, float step_size, float bc2_sqrt) {
    const f4 p4 = ld4(t.p + i), m4 = ld4(t.m + i), v4 = ld4(t.v + i);
    float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
    const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int r = 0; r < 4; ++r)
        adam_one(pp[r], gg[r], mm[r], vv[r], f.decay_eff, f.wd, f.decoupled, f.w1, f.b2, f.w2, step_size, bc2_sqrt, f.eps);
    st4(t.p + i, f4{pp[0], pp[1], pp[2], pp[3]}), st4(t.m + i, f4{mm[0], mm[1], mm[2], mm[3]}), st4(t.v + i, f4{vv[0], vv[1], vv[2], vv[3]});
}

// What the later launches of the step read as tables, written by the first workgroup of launch 1 (both change every step):
//   W2T  [idx = ch * 8 + c][col]: layer 2's weights, classes padded to 8 -- a lane's four columns of one output are one
//        16-byte load, for the forward projection (launch 2) and for dH = dZ2 W2^T (launch 4) alike
//   FACT [layer * 17 + role][2]: the step-dependent Adam factors of every tensor (double-precision pow: once, not per block)
__device__ __forceinline__ void write_tables(const SmallDev& d) {
    if (blockIdx.x != 0) return;
    for (int e = threadIdx.x; e < 3 * F * C8; e += blockDim.x) {
        const int idx = e / F, col = e % F, ch = idx >> 3, c = idx & 7;
        d.W2T[e] = c < d.C ? d.t[1][ACM_SR_W_LOW + ch].p[col * d.C + c] : 0.f;
    }
    if ((int)threadIdx.x < 2 * ACM_SMALL_ROLES && d.update) {
        const SmallTensor& t = d.t[threadIdx.x / ACM_SMALL_ROLES][threadIdx.x % ACM_SMALL_ROLES];
        if (t.p && t.step) acm_adam_step_factors(d.hp, t.step[0], d.FACT[2 * threadIdx.x], d.FACT[2 * threadIdx.x + 1]);
    }
    // the dropout counter of THIS step: the later launches draw their masks with this copy, so the last launch may advance
    // the live counter whenever it likes (no arrival barrier over its ~900 blocks: their atomics on one address were 10 us)
    if (threadIdx.x == 0) d.latch[0] = d.drop_hidden.step ? d.drop_hidden.step[0] : (d.drop_in.step ? d.drop_in.step[0] : 0);
}

// ------------------------------------------------------------------------------------------------ launch 1: Z1 = drop(X) Wcat
__global__ __launch_bounds__(256) void small_proj1_kernel(SmallDev d) {
    write_tables(d);
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    const int wave = (int)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int)gridDim.x * 4;
    const AcmDropCtx dc = acm_drop_ctx(d.drop_in);
    const float* tab[3] = {d.t[0][ACM_SR_W_LOW].p, d.t[0][ACM_SR_W_HIGH].p, d.t[0][ACM_SR_W_MLP].p};
    const int ld[3] = {F, F, F};
    const float* __restrict__ xv = d.x_vals;
    wide_items<3>(
        d.x, d.slots, d.counters, wave, n_waves, [](int) { return 0; },
        [&](const AcmItem& item, f4(&acc)[3]) {
            wide_gather<3, 0>(d.x.indices, item.begin, item.end, tab, ld, 0,
                              [&](int pos) { return xv[pos] * acm_drop1(dc, pos, 0); }, acc);
        },
        [&](int row, f4(&acc)[3], int) {
            if (g < 3) st4(d.Z1 + (long)row * (3 * F) + g * F + 4 * m, sel4(g, acc[0], acc[1], acc[2], acc[2]));
            // the structure parameter of layer 2 as a block of the narrow table (refreshed every step: it is a parameter)
            if (d.k == 4 && g == 3 && m < C8) d.T2[(long)row * 24 + 16 + m] = m < d.C ? d.t[1][ACM_SR_STRUC].p[(long)row * d.C + m] : 0.f;
        });
}

// ------------------------------------------------------------------------------------------------ launch 2: layer 1 forward
// per-row math of a WIDE layer (64 columns), group g = channel g.  `acc` = complete gathered sums (in every group).
struct Pre2 {
    float rs;
    f4 own;
};
template <bool FOUR, bool VARIANT>
__global__ __launch_bounds__(64 * WAVES) void small_conv1_fwd_kernel(SmallDev d, const float* z1) {
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    const int wave = (int)blockIdx.x * WAVES + (threadIdx.x >> 6), n_waves = (int)gridDim.x * WAVES;
    constexpr int NCH = FOUR ? 3 : 2;
    constexpr int K = FOUR ? 4 : 3;
    const AcmDropCtx dh = acm_drop_ctx(d.drop_hidden);
    const float* tab[3] = {z1, z1 + F, FOUR ? d.t[0][ACM_SR_STRUC].p : z1};
    const int ld[3] = {3 * F, 3 * F, F};
    const bool act = g < K;
    // this lane's channel parameters (columns 4m .. 4m+3) and its share of layer 2's weights: output idx = g + 4t
    const f4 av = act ? ld4(d.t[0][ACM_SR_V_LOW + g].p + 4 * m) : zero4();
    f4 gam = zero4(), bet = zero4();
    if (d.layernorm && act) gam = ld4(d.t[0][ACM_SR_LNW_LOW + g].p + 4 * m), bet = ld4(d.t[0][ACM_SR_LNB_LOW + g].p + 4 * m);
    float mix[K][K];
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
        for (int b = 0; b < K; ++b) mix[a][b] = d.t[0][ACM_SR_MIX].p[a * K + b];
    f4 w2t[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) w2t[t] = ld4(d.W2T + (g + 4 * t) * F + 4 * m);
    // the row's own terms: group 1 its Z_H row, group 2 its Z_I row, group 3 its struc_low row
    const float* own_base = g == 1 ? z1 + F : (g == 2 ? z1 + 2 * F : (FOUR ? d.t[0][ACM_SR_STRUC].p : z1));
    const int own_ld = (g == 1 || g == 2) ? 3 * F : F;
    wide_items<NCH>(
        d.graph, d.slots, d.counters, wave, n_waves,
        [&](int row) {
            Pre2 pf;
            pf.rs = d.row_scale[row];
            pf.own = (g == 0 || (!FOUR && g == 3)) ? zero4() : ld4(own_base + (long)row * own_ld + 4 * m);
            return pf;
        },
        [&](const AcmItem& item, f4(&acc)[3]) {
            wide_gather<NCH, VARIANT ? 3 : 0>(d.graph.indices, item.begin, item.end, tab, ld, item.row, [](int) { return 1.f; }, acc);
        },
        [&](int row, f4(&acc)[3], const Pre2& pf) {
            const float rs = pf.rs;
            const f4 own = pf.own;
            f4 h;
            if (VARIANT) h = sel4(g, rs * acc[0], relu4(own) - rs * acc[1], relu4(own), relu4(acc[2] - own));
            else h = sel4(g, relu4(rs * acc[0]), relu4(own - rs * acc[1]), relu4(own), relu4(acc[2] - own));
            if (!act) h = zero4();
            float mean = 0.f, rstd = 1.f;
            f4 hn = h;
            if (d.layernorm) {
                mean = acm_group_sum<16>(hsum4(h)) * (1.0f / F);
                const f4 dd = h - mean;
                const float var = acm_group_sum<16>(hsum4(dd * dd)) * (1.0f / F);
                rstd = 1.0f / sqrtf(var + ACM_LN_EPS);
                hn = dd * rstd * gam + bet;
            }
            const float logit = acm_group_sum<16>(hsum4(hn * av));
            const float sg = 1.0f / (1.0f + expf(-logit));
            float sig[K], tt[K], al[K];
#pragma unroll
            for (int c = 0; c < K; ++c) sig[c] = acm_lane_f(sg, 16 * c);
            float mx = -INFINITY;
#pragma unroll
            for (int b = 0; b < K; ++b) {
                float s = 0.f;
#pragma unroll
                for (int a = 0; a < K; ++a) s += sig[a] * mix[a][b];
                tt[b] = s / (float)K;
                mx = fmaxf(mx, tt[b]);
            }
            float den = 0.f;
#pragma unroll
            for (int b = 0; b < K; ++b) al[b] = expf(tt[b] - mx), den += al[b];
#pragma unroll
            for (int b = 0; b < K; ++b) al[b] /= den;
            const float my_al = g == 0 ? al[0] : (g == 1 ? al[1] : (g == 2 ? al[2] : (K == 4 ? al[K - 1] : 0.f)));
            f4 out = d.scale * xsum4(my_al * h);
            // the caller's dropout(relu(.)) between the layers (models.py:70)
            out = relu4(out);
            if (dh.on) {
                out.x *= acm_drop1(dh, row, 4 * m), out.y *= acm_drop1(dh, row, 4 * m + 1);
                out.z *= acm_drop1(dh, row, 4 * m + 2), out.w *= acm_drop1(dh, row, 4 * m + 3);
            }
            if (g == 0) st4(d.OUT1 + (long)row * F + 4 * m, out);
            if (act) st4(d.H1 + (long)row * (4 * F) + g * F + 4 * m, h);
            if (m == 0 && act) {
                float* st = d.ST1 + (long)row * 16;
                st[g] = mean, st[4 + g] = rstd, st[8 + g] = sg, st[12 + g] = my_al;
                d.att1[(long)row * 4 + g] = my_al;
            }
            // layer 2's projection of this row: output idx = ch * 8 + c, group g takes idx = g, g + 4, ...
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int idx = g + 4 * t, ch = idx >> 3, c = idx & 7;
                const float s = acm_group_sum<16>(hsum4(out * w2t[t]));
                if (m == 0 && c < d.C) {
                    if (ch < 2) d.T2[(long)row * 24 + ch * 8 + c] = s;
                    else d.Z2I[(long)row * C8 + c] = s;
                }
            }
        });
}

// ------------------------------------------------------------------------------------------------ launch 3: layer 2 forward + loss + K3
struct Pre3 {
    float rs, zH, zI, sS, w;
    int y;
};
template <bool FOUR>
__global__ __launch_bounds__(64 * WAVES) void small_conv2_fwd_kernel(SmallDev d) {
    __shared__ float red[WAVES][PART3_PITCH];
    const int lane = threadIdx.x & 63, c = lane & 7, wv = threadIdx.x >> 6;
    const int wave = (int)blockIdx.x * WAVES + wv, n_waves = (int)gridDim.x * WAVES;
    constexpr int NQ = FOUR ? 6 : 4;
    constexpr int K = FOUR ? 4 : 3;
    const int C = d.C;
    const bool valid = c < C;
    const float invC = 1.0f / (float)C;
    const bool variant = d.relu_before != 0;
    float av[K], gam[K], bet[K];
#pragma unroll
    for (int ch = 0; ch < K; ++ch) {
        av[ch] = valid ? d.t[1][ACM_SR_V_LOW + ch].p[c] : 0.f;
        gam[ch] = (valid && d.layernorm) ? d.t[1][ACM_SR_LNW_LOW + ch].p[c] : 0.f;
        bet[ch] = (valid && d.layernorm) ? d.t[1][ACM_SR_LNB_LOW + ch].p[c] : 0.f;
    }
    float mix[K][K];
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
        for (int b = 0; b < K; ++b) mix[a][b] = d.t[1][ACM_SR_MIX].p[a * K + b];
    float a_dv[K], a_dg[K], a_db[K], a_dm[K][K], a_loss = 0.f;
#pragma unroll
    for (int ch = 0; ch < K; ++ch) {
        a_dv[ch] = a_dg[ch] = a_db[ch] = 0.f;
#pragma unroll
        for (int b = 0; b < K; ++b) a_dm[ch][b] = 0.f;
    }
    narrow_items<NQ>(
        d.graph, d.slots, d.counters, wave, n_waves,
        [&](int row) {
            Pre3 pf;
            pf.rs = d.row_scale[row];
            pf.zH = d.T2[(long)row * 24 + 8 + c], pf.zI = d.Z2I[(long)row * C8 + c];
            pf.sS = FOUR ? d.T2[(long)row * 24 + 16 + c] : 0.f;
            pf.w = d.train ? d.row_weight[row] : 0.f;
            pf.y = d.train ? (int)d.labels[row] : 0;
            return pf;
        },
        [&](const AcmItem& item) { return narrow_gather<NQ>(d.graph.indices, item.begin, item.end, d.T2, 24, item.row, variant); },
        [&](int row, f4 acc, const Pre3& pf, int li) {
            const float rs = pf.rs;
            const float aL = narrow_pick(acc, 0), aH = narrow_pick(acc, 1), aS = FOUR ? narrow_pick(acc, 2) : 0.f;
            const float zH = pf.zH, zI = pf.zI;
            float h[K];
            if (variant) h[0] = rs * aL, h[1] = fmaxf(zH, 0.f) - rs * aH;
            else h[0] = fmaxf(rs * aL, 0.f), h[1] = fmaxf(zH - rs * aH, 0.f);
            h[2] = fmaxf(zI, 0.f);
            if (FOUR) h[K - 1] = fmaxf(aS - pf.sS, 0.f);
            float xh[K], hn[K], rstd[K], sig[K], tt[K], al[K];
#pragma unroll
            for (int ch = 0; ch < K; ++ch) {
                if (!valid) h[ch] = 0.f;
                xh[ch] = 0.f, rstd[ch] = 1.f, hn[ch] = h[ch];
                if (d.layernorm) {
                    const float mean = gsum8(h[ch]) * invC;
                    const float dd = valid ? h[ch] - mean : 0.f;
                    const float var = gsum8(dd * dd) * invC;
                    rstd[ch] = 1.0f / sqrtf(var + ACM_LN_EPS);
                    xh[ch] = dd * rstd[ch];
                    hn[ch] = valid ? xh[ch] * gam[ch] + bet[ch] : 0.f;
                }
                const float logit = gsum8(hn[ch] * av[ch]);
                sig[ch] = 1.0f / (1.0f + expf(-logit));
            }
            float mx = -INFINITY;
#pragma unroll
            for (int b = 0; b < K; ++b) {
                float s = 0.f;
#pragma unroll
                for (int a = 0; a < K; ++a) s += sig[a] * mix[a][b];
                tt[b] = s / (float)K;
                mx = fmaxf(mx, tt[b]);
            }
            float den = 0.f;
#pragma unroll
            for (int b = 0; b < K; ++b) al[b] = expf(tt[b] - mx), den += al[b];
            float z = 0.f;
#pragma unroll
            for (int b = 0; b < K; ++b) al[b] /= den, z += al[b] * h[b];
            z *= d.scale;
            if (lane < 8 && valid) d.logits[(long)row * C + c] = z;
            if (lane < K) d.att2[(long)row * 4 + lane] = lane == 0 ? al[0] : (lane == 1 ? al[1] : (lane == 2 ? al[2] : al[K - 1]));
            if (!d.train) return;
            // masked NLL of the row and its gradient (acm_nll_row)
            const float w = pf.w;
            float dl = 0.f, r_loss = 0.f;
            float r_dv[K], r_dg[K], r_db[K], r_dm[K][K];        // this row's terms of the parameter gradients
            if (w != 0.f) {
                const float mz = gmax8(valid ? z : -INFINITY);
                const float ex = valid ? expf(z - mz) : 0.f;
                const float s = gsum8(ex);
                const int y = pf.y;
                const float zy = __shfl(z, (lane & ~7) + y);
                r_loss = w * ((mz + logf(s)) - zy);
                dl = valid ? w * (ex * (1.0f / s) - (c == y ? 1.f : 0.f)) : 0.f;
            }
            // row-local backward of the layer (no post-op on an output layer)
            float dal[K], dot = 0.f;
#pragma unroll
            for (int ch = 0; ch < K; ++ch) dal[ch] = d.scale * gsum8(dl * h[ch]), dot += al[ch] * dal[ch];
            float dt[K];
#pragma unroll
            for (int ch = 0; ch < K; ++ch) dt[ch] = al[ch] * (dal[ch] - dot);
            float gch[K];
#pragma unroll
            for (int ch = 0; ch < K; ++ch) {
                float dsig = 0.f;
#pragma unroll
                for (int b = 0; b < K; ++b) {
                    dsig += dt[b] * mix[ch][b];
                    r_dm[ch][b] = sig[ch] * dt[b] / (float)K;
                }
                dsig /= (float)K;
                const float dlg = dsig * sig[ch] * (1.0f - sig[ch]);
                r_dv[ch] = dlg * hn[ch];
                r_dg[ch] = r_db[ch] = 0.f;
                const float dhn = dlg * av[ch];
                float dln = dhn;
                if (d.layernorm) {
                    r_dg[ch] = dhn * xh[ch];
                    r_db[ch] = valid ? dhn : 0.f;
                    const float u = dhn * gam[ch];
                    const float s1 = gsum8(u) * invC, s2 = gsum8(u * xh[ch]) * invC;
                    dln = valid ? rstd[ch] * (u - s1 - xh[ch] * s2) : 0.f;
                }
                const float dh = d.scale * al[ch] * dl + dln;
                const bool relu_here = !(variant && ch < 2);
                gch[ch] = (relu_here && !(h[ch] > 0.f)) ? 0.f : dh;
            }
            if (lane < 8) {
                float* gp = d.G2 + (long)row * 24;
                gp[c] = rs * gch[0], gp[8 + c] = rs * gch[1];
                if (FOUR) gp[16 + c] = gch[K - 1];
                float* dz = d.DZ2 + (long)row * 24;
                dz[8 + c] = gch[1], dz[16 + c] = gch[2];
            }
            if (li < 0) {
                a_loss += r_loss;
#pragma unroll
                for (int ch = 0; ch < K; ++ch) {
                    a_dv[ch] += r_dv[ch], a_dg[ch] += r_dg[ch], a_db[ch] += r_db[ch];
#pragma unroll
                    for (int b = 0; b < K; ++b) a_dm[ch][b] += r_dm[ch][b];
                }
            } else {                                      // the finisher of a long row: its terms go to the row's own record
                float* rec = d.long3 + (long)li * PART3_PITCH;
                if (lane < 8) {
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) {
                        const bool on = ch < K;
                        rec[ch * 8 + c] = on ? r_dv[on ? ch : 0] : 0.f, rec[32 + ch * 8 + c] = on ? r_dg[on ? ch : 0] : 0.f;
                        rec[64 + ch * 8 + c] = on ? r_db[on ? ch : 0] : 0.f;
                    }
                }
                if (lane == 0) {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) rec[96 + a * 4 + b] = (a < K && b < K) ? r_dm[a < K ? a : 0][b < K ? b : 0] : 0.f;
                    rec[112] = r_loss;
                }
            }
        });
    if (!d.train) return;
    // per-workgroup partial sums: waves in order through LDS (deterministic)
    if (lane < 8) {
#pragma unroll
        for (int ch = 0; ch < K; ++ch) red[wv][ch * 8 + c] = a_dv[ch], red[wv][32 + ch * 8 + c] = a_dg[ch], red[wv][64 + ch * 8 + c] = a_db[ch];
        if (!FOUR) red[wv][24 + c] = red[wv][56 + c] = red[wv][88 + c] = 0.f;
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) red[wv][96 + a * 4 + b] = (a < K && b < K) ? a_dm[a < K ? a : 0][b < K ? b : 0] : 0.f;
        red[wv][112] = a_loss;
    }
    __syncthreads();
    if (threadIdx.x < PART3) {
        float s = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += red[w][threadIdx.x];
        d.part3[(long)blockIdx.x * PART3_PITCH + threadIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------------ launch 4: layer 2 backward gather + dH + layer 1 K3
struct Pre4 {
    float rs, gH, dzI, zL, zH, gS, sp, sm, sv;
    f4 o, h;
    float mean, rstd, sa[8];
};
template <bool FOUR>
__global__ __launch_bounds__(64 * WAVES) void small_conv2_bwd_kernel(SmallDev d, const float* z1) {
    __shared__ __attribute__((aligned(16))) float red[WAVES / 2][PART4];  // the waves' sums side by side (two rounds: 37 KB)
    __shared__ __attribute__((aligned(16))) float w2s[3 * F * C8];        // W2T [idx][col]: 24 rows of 64, read by every row's dH
    for (int e = 4 * threadIdx.x; e < 3 * F * C8; e += 4 * 64 * WAVES) st4(w2s + e, ld4(d.W2T + e));
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15, c = lane & 7, wv = threadIdx.x >> 6;
    const int wave = (int)blockIdx.x * WAVES + wv, n_waves = (int)gridDim.x * WAVES;
    constexpr int NQ = FOUR ? 6 : 4;
    constexpr int K = FOUR ? 4 : 3;
    const int C = d.C;
    const bool variant = d.relu_before != 0;
    const AcmDropCtx dh = acm_drop_ctx(d.drop_hidden);
    const float inv_keep = dh.on ? dh.inv_keep : 1.f;
    const bool act = g < K;
    const f4 av = act ? ld4(d.t[0][ACM_SR_V_LOW + g].p + 4 * m) : zero4();
    f4 gam = zero4();
    if (d.layernorm && act) gam = ld4(d.t[0][ACM_SR_LNW_LOW + g].p + 4 * m);
    f4 bet = zero4();
    if (d.layernorm && act) bet = ld4(d.t[0][ACM_SR_LNB_LOW + g].p + 4 * m);
    float mix[K][K];
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
        for (int b = 0; b < K; ++b) mix[a][b] = d.t[0][ACM_SR_MIX].p[a * K + b];
    const AdamFactors af(d.hp);
    const SmallTensor& ts2 = d.t[1][ACM_SR_STRUC];
    const bool s2_lane = FOUR && lane < 8 && c < C;
    f4 a_w2[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) a_w2[t] = zero4();
    f4 a_dv = zero4(), a_dg = zero4(), a_db = zero4();
    float a_dm = 0.f;                                 // d(mix): lane L < 16 holds element (L >> 2, L & 3)
    narrow_items<NQ>(
        d.graph, d.slots, d.counters, wave, n_waves,
        [&](int row) {                                   // everything the row's epilogue reads that is not a gathered sum
            Pre4 pf;
            pf.rs = d.row_scale[row];
            const float* dzp = d.DZ2 + (long)row * 24;
            pf.gH = dzp[8 + c], pf.dzI = dzp[16 + c];
            pf.zL = variant ? d.T2[(long)row * 24 + c] : 1.f, pf.zH = variant ? d.T2[(long)row * 24 + 8 + c] : 1.f;
            pf.gS = FOUR ? d.G2[(long)row * 24 + 16 + c] : 0.f;
            pf.sp = pf.sm = pf.sv = 0.f;
            if (s2_lane && d.update) {
                const long i = (long)row * C + c;
                pf.sp = ts2.p[i], pf.sm = ts2.m[i], pf.sv = ts2.v[i];
            }
            pf.o = ld4(d.OUT1 + (long)row * F + 4 * m);
            pf.h = act ? ld4(d.H1 + (long)row * (4 * F) + g * F + 4 * m) : zero4();
            const float* st = d.ST1 + (long)row * 16;
            pf.mean = st[g], pf.rstd = st[4 + g];
#pragma unroll
            for (int q = 0; q < 8; ++q) pf.sa[q] = st[8 + q];         // sigmoid and alpha of the four channels (wave-uniform)
            return pf;
        },
        [&](const AcmItem& item) { return narrow_gather<NQ>(d.graph.indices, item.begin, item.end, d.G2, 24, item.row, false); },
        [&](int row, f4 acc, const Pre4& pf, int li) {
            const float aL = narrow_pick(acc, 0), aH = narrow_pick(acc, 1), aS = FOUR ? narrow_pick(acc, 2) : 0.f;
            float dzL = aL, dzH = pf.gH - aH;
            const float dzI = pf.dzI;
            if (variant) {
                if (!(pf.zL > 0.f)) dzL = 0.f;
                if (!(pf.zH > 0.f)) dzH = 0.f;
            }
            if (s2_lane) {                                // dS2 = P G_S - G_S: final here, the parameter row is updated in place
                const float ds = aS - pf.gS;
                const long i = (long)row * C + c;
                if (ts2.g) ts2.g[i] = ds;
                if (d.update) {
                    float pp = pf.sp, mm = pf.sm, vv = pf.sv;
                    const float* fc = d.FACT + 2 * (ACM_SMALL_ROLES + ACM_SR_STRUC);
                    adam_one(pp, ds, mm, vv, af.decay_eff, af.wd, af.decoupled, af.w1, af.b2, af.w2, fc[0], fc[1], af.eps);
                    ts2.p[i] = pp, ts2.m[i] = mm, ts2.v[i] = vv;
                }
            }
            // dZ2 of this row as wave-uniform values, idx = ch * 8 + c
            float dz[24];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                dz[cc] = acm_lane_f(dzL, cc);
                dz[8 + cc] = acm_lane_f(dzH, cc);
                dz[16 + cc] = acm_lane_f(dzI, cc);
            }
            const f4 o = pf.o;
            // dH = dZ2 Wcat2^T (columns 4m .. 4m+3) and dWcat2 += H^T dZ2 (group g: idx = g, g + 4, ...)
            // (W2T from LDS, staged once per workgroup: as global loads its 24 rows were three dependent round trips per row)
            f4 dH0 = zero4(), dH1 = zero4();
#pragma unroll
            for (int q = 0; q < 24; q += 2) dH0 += dz[q] * ld4(w2s + q * F + 4 * m), dH1 += dz[q + 1] * ld4(w2s + (q + 1) * F + 4 * m);
            const f4 dH = dH0 + dH1;
            if (li < 0) {
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const float zz = g == 0 ? dz[4 * t] : (g == 1 ? dz[4 * t + 1] : (g == 2 ? dz[4 * t + 2] : dz[4 * t + 3]));
                    a_w2[t] += zz * o;
                }
            } else {                                      // a long row: its dW2 term goes to the row's own record
                float* rec = d.long4 + (long)li * LONG4;
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const int idx = g + 4 * t, ch = idx >> 3, cc = idx & 7;
                    const float zz = g == 0 ? dz[4 * t] : (g == 1 ? dz[4 * t + 1] : (g == 2 ? dz[4 * t + 2] : dz[4 * t + 3]));
                    float* rp = rec + ch * (F * C8) + (4 * m) * C8 + cc;
                    rp[0] = zz * o.x, rp[C8] = zz * o.y, rp[2 * C8] = zz * o.z, rp[3 * C8] = zz * o.w;
                }
            }
            // through dropout(relu(.)): the forward's output is non-zero exactly where both let the element pass
            f4 dmix;
            dmix.x = o.x > 0.f ? dH.x * inv_keep : 0.f, dmix.y = o.y > 0.f ? dH.y * inv_keep : 0.f;
            dmix.z = o.z > 0.f ? dH.z * inv_keep : 0.f, dmix.w = o.w > 0.f ? dH.w * inv_keep : 0.f;
            // row-local backward of layer 1, channel g in group g
            const float rs = pf.rs;
            const f4 h = pf.h;
            float sig[K], al[K];
#pragma unroll
            for (int ch = 0; ch < K; ++ch) sig[ch] = pf.sa[ch], al[ch] = pf.sa[4 + ch];
            const float mean = pf.mean, rstd = pf.rstd;
            const float dal_mine = d.scale * acm_group_sum<16>(hsum4(dmix * h));
            float dal[K], dot = 0.f;
#pragma unroll
            for (int ch = 0; ch < K; ++ch) dal[ch] = acm_lane_f(dal_mine, 16 * ch), dot += al[ch] * dal[ch];
            float dt[K], dlg[K];
#pragma unroll
            for (int ch = 0; ch < K; ++ch) dt[ch] = al[ch] * (dal[ch] - dot);
#pragma unroll
            for (int ch = 0; ch < K; ++ch) {
                float dsig = 0.f;
#pragma unroll
                for (int b = 0; b < K; ++b) dsig += dt[b] * mix[ch][b];
                dsig /= (float)K;
                dlg[ch] = dsig * sig[ch] * (1.0f - sig[ch]);
            }
            float r_dm = 0.f;
            {
                const int ma = (lane >> 2) & 3, mb = lane & 3;
                const float sa = ma == 0 ? sig[0] : (ma == 1 ? sig[1] : (ma == 2 ? sig[2] : sig[K - 1]));
                const float tb = mb == 0 ? dt[0] : (mb == 1 ? dt[1] : (mb == 2 ? dt[2] : dt[K - 1]));
                if (ma < K && mb < K) r_dm = sa * tb / (float)K;
            }
            const float my_dl = g == 0 ? dlg[0] : (g == 1 ? dlg[1] : (g == 2 ? dlg[2] : (K == 4 ? dlg[K - 1] : 0.f)));
            const float my_al = g == 0 ? al[0] : (g == 1 ? al[1] : (g == 2 ? al[2] : (K == 4 ? al[K - 1] : 0.f)));
            f4 xh = zero4(), hn = h;
            if (d.layernorm) xh = (h - mean) * rstd, hn = xh * gam + bet;
            const f4 r_dv = my_dl * hn;
            const f4 dhn = my_dl * av;
            f4 dln = dhn, r_dg = zero4(), r_db = zero4();
            if (d.layernorm) {
                r_dg = dhn * xh;
                r_db = dhn;
                const f4 u = dhn * gam;
                const float s1 = acm_group_sum<16>(hsum4(u)) * (1.0f / F), s2 = acm_group_sum<16>(hsum4(u * xh)) * (1.0f / F);
                dln = rstd * (u - s1 - xh * s2);
            }
            f4 dhc = (d.scale * my_al) * dmix + dln;
            const bool relu_here = !(variant && g < 2);
            if (relu_here) {
                dhc.x = h.x > 0.f ? dhc.x : 0.f, dhc.y = h.y > 0.f ? dhc.y : 0.f;
                dhc.z = h.z > 0.f ? dhc.z : 0.f, dhc.w = h.w > 0.f ? dhc.w : 0.f;
            }
            float* g1 = d.G1 + (long)row * (3 * F) + 4 * m;
            float* dz1 = d.DZ1 + (long)row * (3 * F) + 4 * m;
            if (g == 0) st4(g1, rs * dhc);
            if (g == 1) st4(g1 + F, rs * dhc), st4(dz1 + F, dhc);
            if (g == 2) st4(dz1 + 2 * F, dhc);
            if (FOUR && g == 3) st4(g1 + 2 * F, dhc);
            if (li < 0) {
                a_dv += r_dv, a_dg += r_dg, a_db += r_db, a_dm += r_dm;
            } else {                                      // the finisher of a long row: its terms go to the row's own record
                float* rec = d.long4 + (long)li * LONG4 + 3 * F * C8;
                const f4 zv = zero4();
                st4(rec + g * F + 4 * m, act ? r_dv : zv), st4(rec + 4 * F + g * F + 4 * m, act ? r_dg : zv);
                st4(rec + 8 * F + g * F + 4 * m, act ? r_db : zv);
                if (lane < 16) rec[12 * F + lane] = r_dm;
            }
        });
    // per-workgroup partial sums: waves 0-3 leave their sums in a slice each, waves 4-7 add theirs on top, then every thread
    // adds the four slices of its elements -- a fixed order of additions (deterministic), two barriers
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if ((wv >> 2) == round) {
            float* mine = red[wv & 3];
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int idx = g + 4 * t, ch = idx >> 3, cc = idx & 7;
                float* rp = mine + ch * (F * C8) + (4 * m) * C8 + cc;
                if (round == 0) rp[0] = a_w2[t].x, rp[C8] = a_w2[t].y, rp[2 * C8] = a_w2[t].z, rp[3 * C8] = a_w2[t].w;
                else rp[0] += a_w2[t].x, rp[C8] += a_w2[t].y, rp[2 * C8] += a_w2[t].z, rp[3 * C8] += a_w2[t].w;
            }
            float* r1 = mine + 3 * F * C8 + g * F + 4 * m;
            const f4 zv = zero4();
            const f4 v0 = act ? a_dv : zv, v1 = act ? a_dg : zv, v2 = act ? a_db : zv;
            if (round == 0) st4(r1, v0), st4(r1 + 4 * F, v1), st4(r1 + 8 * F, v2);
            else st4(r1, ld4(r1) + v0), st4(r1 + 4 * F, ld4(r1 + 4 * F) + v1), st4(r1 + 8 * F, ld4(r1 + 8 * F) + v2);
            if (lane < 16) {
                float* rm = mine + 3 * F * C8 + 12 * F + lane;
                if (round == 0) rm[0] = a_dm;
                else rm[0] += a_dm;
            }
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < PART4; e += 64 * WAVES)
        d.part4[(long)blockIdx.x * PART4 + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// ------------------------------------------------------------------------------------------------ launch 5: layer 1 backward gather
struct Pre5 {
    f4 a, b, pp, mm, vv;
};
template <bool FOUR, bool VARIANT>
__global__ __launch_bounds__(64 * WAVES) void small_conv1_bwd_kernel(SmallDev d, const float* z1) {
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    const int wave = (int)blockIdx.x * WAVES + (threadIdx.x >> 6), n_waves = (int)gridDim.x * WAVES;
    constexpr int NCH = FOUR ? 3 : 2;
    const float* tab[3] = {d.G1, d.G1 + F, d.G1 + 2 * F};
    const int ld[3] = {3 * F, 3 * F, 3 * F};
    const AdamFactors af(d.hp);
    const SmallTensor& ts = d.t[0][ACM_SR_STRUC];
    wide_items<NCH>(
        d.graph, d.slots, d.counters, wave, n_waves,
        [&](int row) {
            // group 0: the ReLU mask of Z_L (ACMII); group 1: its direct term dZ_H and the mask of Z_H; group 3: G_S of the row
            // and the parameter row with its moments
            Pre5 pf;
            pf.a = pf.b = pf.pp = pf.mm = pf.vv = zero4();
            const float* zr = z1 + (long)row * (3 * F) + 4 * m;
            if (g == 0 && VARIANT) pf.b = ld4(zr);
            if (g == 1) {
                pf.a = ld4(d.DZ1 + (long)row * (3 * F) + F + 4 * m);
                if (VARIANT) pf.b = ld4(zr + F);
            }
            if (FOUR && g == 3) {
                pf.a = ld4(d.G1 + (long)row * (3 * F) + 2 * F + 4 * m);
                if (d.update) {
                    const long i = (long)row * F + 4 * m;
                    pf.pp = ld4(ts.p + i), pf.mm = ld4(ts.m + i), pf.vv = ld4(ts.v + i);
                }
            }
            return pf;
        },
        [&](const AcmItem& item, f4(&acc)[3]) {
            wide_gather<NCH, 0>(d.graph.indices, item.begin, item.end, tab, ld, item.row, [](int) { return 1.f; }, acc);
        },
        [&](int row, f4(&acc)[3], const Pre5& pf) {
            float* dz1 = d.DZ1 + (long)row * (3 * F) + 4 * m;
            if (g == 0 || g == 1) {
                f4 dz = g == 0 ? acc[0] : pf.a - acc[1];
                if (VARIANT) {
                    const f4 z = pf.b;
                    dz.x = z.x > 0.f ? dz.x : 0.f, dz.y = z.y > 0.f ? dz.y : 0.f, dz.z = z.z > 0.f ? dz.z : 0.f, dz.w = z.w > 0.f ? dz.w : 0.f;
                }
                st4(dz1 + g * F, dz);
            } else if (FOUR && g == 3) {                   // dS = P G_S - G_S: final, the parameter row is updated in place
                const f4 ds = acc[2] - pf.a;
                const long i = (long)row * F + 4 * m;
                if (ts.g) st4(ts.g + i, ds);
                if (d.update) {
                    const float* fc = d.FACT + 2 * ACM_SR_STRUC;
                    float pp[4] = {pf.pp.x, pf.pp.y, pf.pp.z, pf.pp.w}, mm[4] = {pf.mm.x, pf.mm.y, pf.mm.z, pf.mm.w};
                    float vv[4] = {pf.vv.x, pf.vv.y, pf.vv.z, pf.vv.w};
                    const float gg[4] = {ds.x, ds.y, ds.z, ds.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        adam_one(pp[r], gg[r], mm[r], vv[r], af.decay_eff, af.wd, af.decoupled, af.w1, af.b2, af.w2, fc[0], fc[1], af.eps);
                    st4(ts.p + i, f4{pp[0], pp[1], pp[2], pp[3]}), st4(ts.m + i, f4{mm[0], mm[1], mm[2], mm[3]});
                    st4(ts.v + i, f4{vv[0], vv[1], vv[2], vv[3]});
                }
            }
        });
}

// ------------------------------------------------------------------------------------------------ launch 6: dW1, every other sum, the updates
// blocks [0, item_blocks): dW1 = drop(X)^T dZ1 by feature row (the transposed feature handle's items) + its update;
// blocks behind: RED_EL elements of the two partial-sum vectors per block (sum over the producer workgroups and the long
// rows' records, then the update).  The first block advances the step counters.
__global__ __launch_bounds__(256) void small_finish_kernel(SmallDev d, int item_blocks, int red_blocks) {
    __shared__ float red8[256 / RED_EL][RED_EL];
    const float* __restrict__ fact = d.FACT;              // step factors of every tensor (launch 1 wrote them)
    const AdamFactors af(d.hp);
    auto apply = [&](int layer, int role, long i, float gsum) {
        const SmallTensor& t = d.t[layer][role];
        if (!t.p) return;
        if (t.g) t.g[i] = gsum;
        if (!d.update) return;
        const float* s = fact + 2 * (layer * ACM_SMALL_ROLES + role);
        float pp = t.p[i], mm = t.m[i], vv = t.v[i];
        adam_one(pp, gsum, mm, vv, af.decay_eff, af.wd, af.decoupled, af.w1, af.b2, af.w2, s[0], s[1], af.eps);
        t.p[i] = pp, t.m[i] = mm, t.v[i] = vv;
    };
    const int blk = (int)blockIdx.x;
    if (blk < item_blocks) {
        const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
        const int wave = blk * 4 + (threadIdx.x >> 6), n_waves = item_blocks * 4;
        const AcmDropCtx dc = acm_drop_ctx(d.drop_in);
        const float* tab[3] = {d.DZ1, d.DZ1 + F, d.DZ1 + 2 * F};
        const int ld[3] = {3 * F, 3 * F, 3 * F};
        const float* __restrict__ xv = d.x_vals;
        const float* __restrict__ xtv = d.xt_vals;
        const int32_t* __restrict__ sp = d.xt_src_pos;
        wide_items<3>(
            d.xt, d.slots, d.counters, wave, n_waves,
            [&](int frow) {                                // the weight rows this wave will update, with their moments
                Pre5 pf;
                pf.a = pf.b = pf.pp = pf.mm = pf.vv = zero4();
                if (g < 3 && d.update) {
                    const SmallTensor& t = d.t[0][ACM_SR_W_LOW + g];
                    const long i = (long)frow * F + 4 * m;
                    pf.pp = ld4(t.p + i), pf.mm = ld4(t.m + i), pf.vv = ld4(t.v + i);
                }
                return pf;
            },
            [&](const AcmItem& item, f4(&acc)[3]) {
                if (xtv)
                    wide_gather<3, 0>(d.xt.indices, item.begin, item.end, tab, ld, 0,
                                      [&](int pos) { return xtv[pos] * acm_drop1(dc, sp[pos], 0); }, acc);
                else
                    wide_gather<3, 0>(d.xt.indices, item.begin, item.end, tab, ld, 0,
                                      [&](int pos) { const int s = sp[pos]; return xv[s] * acm_drop1(dc, s, 0); }, acc);
            },
            [&](int frow, f4(&acc)[3], const Pre5& pf) {
                if (g < 3) {
                    const f4 gw = sel4(g, acc[0], acc[1], acc[2], acc[2]);
                    const SmallTensor& t = d.t[0][ACM_SR_W_LOW + g];
                    const long i = (long)frow * F + 4 * m;
                    if (t.g) st4(t.g + i, gw);
                    if (d.update) {
                        const float* fc = fact + 2 * (ACM_SR_W_LOW + g);
                        float pp[4] = {pf.pp.x, pf.pp.y, pf.pp.z, pf.pp.w}, mm[4] = {pf.mm.x, pf.mm.y, pf.mm.z, pf.mm.w};
                        float vv[4] = {pf.vv.x, pf.vv.y, pf.vv.z, pf.vv.w};
                        const float gg[4] = {gw.x, gw.y, gw.z, gw.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            adam_one(pp[r], gg[r], mm[r], vv[r], af.decay_eff, af.wd, af.decoupled, af.w1, af.b2, af.w2, fc[0], fc[1], af.eps);
                        st4(t.p + i, f4{pp[0], pp[1], pp[2], pp[3]}), st4(t.m + i, f4{mm[0], mm[1], mm[2], mm[3]});
                        st4(t.v + i, f4{vv[0], vv[1], vv[2], vv[3]});
                    }
                }
            });
    } else if (blk < item_blocks + red_blocks) {
        // RED_EL elements per block: thread (wl, el) adds the producer workgroups wl, wl + 32, ... (independent loads in
        // flight), the 32 partial sums meet in LDS in a fixed order
        const int el = (int)threadIdx.x & (RED_EL - 1), wl = (int)threadIdx.x / RED_EL;
        const int e = (blk - item_blocks) * RED_EL + el;
        const int K = d.k, C = d.C;
        const bool four = e < PART4;
        const int q = e - PART4;
        float s = 0.f;
        if (e < PART4 + PART3) {
            const float* src = four ? d.part4 + e : d.part3 + q;
            const long pitch = four ? PART4 : PART3_PITCH;
            const int nw = four ? d.nwg4 : d.nwg3;
            // sixteen loads in flight per thread, added in the order of the producer workgroups
            // the long rows' own records follow the workgroups' partials as further producers (their finisher varies from
            // run to run, their place in this sum does not)
            const float* lsrc = four ? d.long4 + e : d.long3 + q;
            const long lpitch = four ? LONG4 : PART3_PITCH;
            const int total = nw + d.n_long;
            constexpr int WL = 256 / RED_EL;
            for (int w0 = wl; w0 < total; w0 += 16 * WL) {
                float v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {             // unconditional loads (a guarded load is waited for before the next)
                    const int w = min(w0 + WL * j, total - 1);
                    const float* pw = w < nw ? src + (long)w * pitch : lsrc + (long)(w - nw) * lpitch;
                    v[j] = *pw;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) s += (w0 + WL * j < total) ? v[j] : 0.f;
            }
        }
        red8[wl][el] = s;
        __syncthreads();
        if (wl == 0 && e < PART4 + PART3) {
            s = 0.f;
#pragma unroll
            for (int j = 0; j < 256 / RED_EL; ++j) s += red8[j][el];
            if (four) {
                if (e < 3 * F * C8) {
                    const int ch = e / (F * C8), col = (e / C8) % F, c = e % C8;
                    if (c < C) apply(1, ACM_SR_W_LOW + ch, (long)col * C + c, s);
                } else if (e < 3 * F * C8 + 12 * F) {
                    const int r = e - 3 * F * C8, kind = r / (4 * F), ch = (r / F) % 4, col = r % F;
                    if (ch < K && (kind == 0 || d.layernorm))
                        apply(0, (kind == 0 ? ACM_SR_V_LOW : (kind == 1 ? ACM_SR_LNW_LOW : ACM_SR_LNB_LOW)) + ch, col, s);
                } else {
                    const int r = e - 3 * F * C8 - 12 * F, a = r / 4, b = r % 4;
                    if (a < K && b < K) apply(0, ACM_SR_MIX, a * K + b, s);
                }
            } else if (q < 96) {
                const int kind = q / 32, ch = (q / 8) % 4, c = q % 8;
                if (ch < K && c < C && (kind == 0 || d.layernorm))
                    apply(1, (kind == 0 ? ACM_SR_V_LOW : (kind == 1 ? ACM_SR_LNW_LOW : ACM_SR_LNB_LOW)) + ch, c, s);
            } else if (q < 112) {
                const int a = (q - 96) / 4, b = (q - 96) % 4;
                if (a < K && b < K) apply(1, ACM_SR_MIX, a * K + b, s);
            } else {
                d.loss[0] = s;
            }
        }
    }
    // nothing in this launch reads a step counter (launch 1 left the factors and the dropout counter of the step in the
    // workspace): the first block advances them
    if (blk == 0 && d.update) {
        if ((int)threadIdx.x < 2 * ACM_SMALL_ROLES) {
            const SmallTensor& t = d.t[threadIdx.x / ACM_SMALL_ROLES][threadIdx.x % ACM_SMALL_ROLES];
            if (t.p && t.step) {
                // tensors that share one step scalar are advanced once (the first role that names it)
                bool first = true;
                for (int o = 0; o < (int)threadIdx.x; ++o) {
                    const SmallTensor& u = d.t[o / ACM_SMALL_ROLES][o % ACM_SMALL_ROLES];
                    if (u.p && u.step == t.step) first = false;
                }
                if (first) t.step[0] += 1.0f;
            }
        }
        if (threadIdx.x == 0 && d.also_advance) d.also_advance[0] += 1;
    }
}

ItemView view_of(const acm_csr* a) {
    ItemView v;
    v.items = a ? a->items : nullptr;
    v.n_items = a ? (int)a->n_items : 0;
    v.long_rows = a ? a->long_rows : nullptr;
    v.long_index = a ? a->long_index : nullptr;
    v.indices = a ? a->indices : nullptr;
    return v;
}

struct Layout {
    size_t Z1, H1, ST1, OUT1, T2, Z2I, G2, DZ2, G1, DZ1, slots, part3, part4, W2T, FACT, latch, long3, long4, counters, total;
};

Layout layout_of(const acm_csr* a, const acm_csr* x, const acm_csr* xt) {
    const size_t n = (size_t)a->n_rows;
    // the slot buffer and the arrival counters serve one launch at a time: sized for the handle with the most pieces
    size_t n_slots = (size_t)a->n_slots, n_long = (size_t)a->n_long;
    if (x) n_slots = std::max(n_slots, (size_t)x->n_slots), n_long = std::max(n_long, (size_t)x->n_long);
    if (xt) n_slots = std::max(n_slots, (size_t)xt->n_slots), n_long = std::max(n_long, (size_t)xt->n_long);
    Layout L;
    size_t o = 0;
    auto take = [&](size_t floats) {
        const size_t at = o;
        o += (floats + 63) / 64 * 64;          // 256-byte aligned blocks
        return at;
    };
    L.Z1 = take(n * 192), L.H1 = take(n * 256), L.ST1 = take(n * 16), L.OUT1 = take(n * 64);
    L.T2 = take(n * 24), L.Z2I = take(n * 8), L.G2 = take(n * 24), L.DZ2 = take(n * 24);
    L.G1 = take(n * 192), L.DZ1 = take(n * 192);
    L.slots = take((n_slots + 1) * 192);
    const size_t wg = (size_t)std::min<int64_t>(MAX_WG, (a->n_items + WAVES - 1) / WAVES);      // = the gather phases' grid
    L.part3 = take(wg * PART3_PITCH), L.part4 = take(wg * PART4);
    L.W2T = take(3 * F * C8), L.FACT = take(2 * ACM_SMALL_ROLES * 2), L.latch = take(4);
    L.long3 = take((size_t)a->n_long * PART3_PITCH + 64), L.long4 = take((size_t)a->n_long * LONG4 + 64);
    L.counters = take(n_long + 64);
    L.total = o * sizeof(float);
    return L;
}

}  // namespace

extern "C" int acm_small_step_workspace_bytes(const acm_csr_t* a_low, const acm_csr_t* x, const acm_csr_t* x_t, size_t* bytes) {
    ACM_REQUIRE(a_low && bytes, ACM_EINVAL, "acm_small_step_workspace_bytes: NULL argument");
    *bytes = layout_of(a_low, x, x_t).total;
    return ACM_OK;
}

extern "C" int acm_small_step(const acm_csr_t* a, const acm_csr_t* x, const acm_csr_t* xt, const acm_small_step_t* p,
                              acm_stream_t stream) {
    ACM_REQUIRE(a && p, ACM_EINVAL, "acm_small_step: NULL argument");
    ACM_REQUIRE(a->vals == nullptr && a->n_rows == a->n_cols, ACM_EUNSUPPORTED,
                "acm_small_step: pattern-only square operators only (acm_csr_create with vals = NULL)");
    ACM_REQUIRE(a->n_rows >= 1 && a->n_rows <= 16384, ACM_EUNSUPPORTED, "acm_small_step: 1 .. 16384 rows (got %lld)", (long long)a->n_rows);
    ACM_REQUIRE(p->n_classes >= 1 && p->n_classes <= C8, ACM_EUNSUPPORTED, "acm_small_step: 1 .. 8 classes (got %d)", p->n_classes);
    ACM_REQUIRE(p->n_channels == 3 || p->n_channels == 4, ACM_EINVAL, "acm_small_step: n_channels must be 3 or 4");
    ACM_REQUIRE(p->row_scale && p->logits && p->att1 && p->att2 && p->workspace, ACM_EINVAL, "acm_small_step: NULL buffer");
    ACM_REQUIRE(x && p->x_vals, ACM_EINVAL, "acm_small_step: the CSR feature handle and its values (x, x_vals)");
    ACM_REQUIRE(x->n_rows == a->n_rows && x->n_cols == p->f_in, ACM_ESHAPE, "acm_small_step: feature handle is %lld x %lld, expected %lld x %d",
                (long long)x->n_rows, (long long)x->n_cols, (long long)a->n_rows, p->f_in);
    const int K = p->n_channels;
    if (p->train) {
        ACM_REQUIRE(p->labels && p->row_weight && p->loss, ACM_EINVAL, "acm_small_step: labels / row_weight / loss");
        ACM_REQUIRE(xt && p->xt_src_pos && xt->n_rows == p->f_in && xt->n_cols == a->n_rows, ACM_EINVAL,
                    "acm_small_step: the transposed feature handle (x_t, xt_src_pos) is needed for dW1");
    }
    const Layout L = layout_of(a, x, p->train ? xt : nullptr);
    ACM_REQUIRE(p->workspace_bytes >= L.total, ACM_ESHAPE, "acm_small_step: workspace of %zu bytes, %zu needed", p->workspace_bytes, L.total);
    // required roles
    for (int l = 0; l < 2; ++l) {
        for (int r = ACM_SR_W_LOW; r <= ACM_SR_V_MLP; ++r)
            ACM_REQUIRE(p->t[l][r].param, ACM_EINVAL, "acm_small_step: layer %d role %d has no parameter", l, r);
        ACM_REQUIRE(p->t[l][ACM_SR_MIX].param, ACM_EINVAL, "acm_small_step: layer %d has no mixing matrix", l);
        if (K == 4)
            ACM_REQUIRE(p->t[l][ACM_SR_V_STRUC].param && p->t[l][ACM_SR_STRUC].param, ACM_EINVAL, "acm_small_step: layer %d lacks the structure channel's parameters", l);
        if (p->layernorm)
            for (int c = 0; c < K; ++c)
                ACM_REQUIRE(p->t[l][ACM_SR_LNW_LOW + c].param && p->t[l][ACM_SR_LNB_LOW + c].param, ACM_EINVAL, "acm_small_step: layer %d lacks LayerNorm parameters", l);
    }
    SmallDev d{};
    d.n = (int)a->n_rows, d.f_in = p->f_in, d.C = p->n_classes, d.k = K;
    d.relu_before = p->relu_before, d.layernorm = p->layernorm, d.train = p->train, d.update = p->update;
    d.scale = p->scale;
    for (int l = 0; l < 2; ++l)
        for (int r = 0; r < ACM_SMALL_ROLES; ++r) {
            const acm_adam_tensor_t& s = p->t[l][r];
            SmallTensor& t = d.t[l][r];
            const bool used = s.param && (r < ACM_SR_LNW_LOW || r == ACM_SR_MIX || r == ACM_SR_STRUC ? true : p->layernorm != 0) &&
                              !((r == ACM_SR_V_STRUC || r == ACM_SR_LNW_STRUC || r == ACM_SR_LNB_STRUC || r == ACM_SR_STRUC) && K == 3);
            if (!used) continue;
            t.p = s.param, t.g = const_cast<float*>(s.grad), t.m = s.exp_avg, t.v = s.exp_avg_sq, t.step = s.step;
            if (p->train && p->update) ACM_REQUIRE(t.m && t.v && t.step, ACM_EINVAL, "acm_small_step: layer %d role %d lacks Adam state", l, r);
            if (p->train && !p->update) ACM_REQUIRE(t.g, ACM_EINVAL, "acm_small_step: layer %d role %d has no gradient buffer (update = 0)", l, r);
        }
    d.graph = view_of(a), d.x = view_of(x), d.xt = view_of(p->train ? xt : nullptr);
    d.x_vals = p->x_vals, d.xt_src_pos = p->xt_src_pos, d.row_scale = p->row_scale;
    d.labels = p->labels, d.row_weight = p->row_weight, d.loss = p->loss, d.logits = p->logits, d.att1 = p->att1, d.att2 = p->att2;
    float* ws = (float*)p->workspace;
    d.Z1 = ws + L.Z1, d.H1 = ws + L.H1, d.ST1 = ws + L.ST1, d.OUT1 = ws + L.OUT1, d.T2 = ws + L.T2, d.Z2I = ws + L.Z2I;
    d.G2 = ws + L.G2, d.DZ2 = ws + L.DZ2, d.G1 = ws + L.G1, d.DZ1 = ws + L.DZ1;
    d.slots = ws + L.slots, d.part3 = ws + L.part3, d.part4 = ws + L.part4, d.W2T = ws + L.W2T, d.FACT = ws + L.FACT;
    d.xt_vals = p->xt_vals;
    d.long3 = ws + L.long3, d.long4 = ws + L.long4, d.n_long = (int)a->n_long;
    d.latch = reinterpret_cast<int64_t*>(ws + L.latch);
    d.counters = (int*)(ws + L.counters);
    d.drop_in = p->drop_in, d.drop_hidden = p->drop_hidden;
    d.hp = AdamScalars{p->lr, p->beta1, p->beta2, p->eps, p->weight_decay, p->decoupled};
    d.also_advance = p->also_advance, d.arrive = p->arrive;
    const float* z1 = d.Z1;
    hipStream_t s = (hipStream_t)stream;
    SmallDev d1 = d;                               // launch 1 reads the live dropout counter and latches it for the others
    if (d.drop_in.p > 0.f) d.drop_in.step = d.latch;
    if (d.drop_hidden.p > 0.f) d.drop_hidden.step = d.latch;
    const int graph_wg = (int)std::min<int64_t>(MAX_WG, (a->n_items + WAVES - 1) / WAVES);
    const int wide_wg = (int)std::min<int64_t>(MAX_WG_WIDE, (a->n_items + WAVES - 1) / WAVES);
    d.nwg3 = d.nwg4 = graph_wg;
    const bool four = K == 4, variant = p->relu_before != 0;
    {   // launch 1
        const int wg = (int)std::min<int64_t>(2 * MAX_WG, (x->n_items + 3) / 4);
        hipLaunchKernelGGL(small_proj1_kernel, dim3(std::max(wg, 1)), dim3(256), 0, s, d1);
    }
    // launch 2
    if (four && variant) hipLaunchKernelGGL((small_conv1_fwd_kernel<true, true>), dim3(wide_wg), dim3(64 * WAVES), 0, s, d, z1);
    else if (four) hipLaunchKernelGGL((small_conv1_fwd_kernel<true, false>), dim3(wide_wg), dim3(64 * WAVES), 0, s, d, z1);
    else if (variant) hipLaunchKernelGGL((small_conv1_fwd_kernel<false, true>), dim3(wide_wg), dim3(64 * WAVES), 0, s, d, z1);
    else hipLaunchKernelGGL((small_conv1_fwd_kernel<false, false>), dim3(wide_wg), dim3(64 * WAVES), 0, s, d, z1);
    // launch 3
    if (four) hipLaunchKernelGGL(small_conv2_fwd_kernel<true>, dim3(graph_wg), dim3(64 * WAVES), 0, s, d);
    else hipLaunchKernelGGL(small_conv2_fwd_kernel<false>, dim3(graph_wg), dim3(64 * WAVES), 0, s, d);
    if (p->train) {
        // launch 4
        if (four) hipLaunchKernelGGL(small_conv2_bwd_kernel<true>, dim3(graph_wg), dim3(64 * WAVES), 0, s, d, z1);
        else hipLaunchKernelGGL(small_conv2_bwd_kernel<false>, dim3(graph_wg), dim3(64 * WAVES), 0, s, d, z1);
        // launch 5
        if (four && variant) hipLaunchKernelGGL((small_conv1_bwd_kernel<true, true>), dim3(wide_wg), dim3(64 * WAVES), 0, s, d, z1);
        else if (four) hipLaunchKernelGGL((small_conv1_bwd_kernel<true, false>), dim3(wide_wg), dim3(64 * WAVES), 0, s, d, z1);
        else if (variant) hipLaunchKernelGGL((small_conv1_bwd_kernel<false, true>), dim3(wide_wg), dim3(64 * WAVES), 0, s, d, z1);
        else hipLaunchKernelGGL((small_conv1_bwd_kernel<false, false>), dim3(wide_wg), dim3(64 * WAVES), 0, s, d, z1);
        // launch 6
        const int item_blocks = (int)std::min<int64_t>(2 * MAX_WG, (xt->n_items + 3) / 4);
        const int red_blocks = (PART4 + PART3 + RED_EL - 1) / RED_EL;
        hipLaunchKernelGGL(small_finish_kernel, dim3(item_blocks + red_blocks), dim3(256), 0, s, d, item_blocks, red_blocks);
    }
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}
