// ACMII first layers with a narrow input (F_in <= 8 < F = 64) on the bf16 matrix pipe, at fp32 accuracy: the mask form.
//
// ACMII (ACM-Geometric/layers.py:94-99, the default of ACM-Geometric: parse.py:57 --variant 1) puts the ReLU BETWEEN
// projection and filter: H_L = A_low relu(X W_L), H_H = relu(X W_H) - A_low relu(X W_H).  acm_conv_acmii.hip gathers the
// neighbour's input row and recomputes relu(x_j [W_L | W_H]) per edge on the fp32 matrix pipe (28 GFLOP per pass: 380 us on
// the twitch-shaped graph), and its backward gathers two 64-wide gradient tables over the transposed operator (7 GB: 600 us).
// Both are replaced here by ONE observation: with the mask m_j[c] = [x_j W[:, c] > 0],
//
//     relu(x_j W[:, c]) = m_j[c] * sum_f x_j[f] W[f, c]
//  => sum_j a_ij relu(x_j W[:, c]) = sum_f W[f, c] * V_i[c, f],      V_i[c, f] = sum_j a_ij m_j[c] x_j[f]
//
// and V_i = (masks of the neighbours)^T (inputs of the neighbours) is a matrix product over the NEIGHBOUR index whose one
// operand is 0 / 1 -- exact in bf16 -- and whose other operand, an fp32 input, is EXACTLY the sum of three bf16 numbers
// (hi + mid + lo, acm_gemm_bx3.hip).  Every product is exact and the fp32 accumulator of v_mfma_f32_16x16x32_bf16 adds
// them: V_i carries the rounding of an fp32 sum over the neighbours, like the reference's spmm, at sixteen times the rate of
// the fp32 MFMA.  The backward needs no transposed product at all (the layer input needs no gradient):
//
//     dW_L[f, c] = sum_i G_L[i, c] * rs_i V^L_i[c, f]            (G = dH, rs_i = 1 / d_i: pattern-only operator)
//     dW_H[f, c] = sum_i G_H[i, c] * (m^H_i[c] x_i[f] - rs_i V^H_i[c, f])
//
// -- the SAME products over the SAME rows, contracted with G instead of W.
//
//   table   one 64-byte row per node, rebuilt every step by acmii_table_kernel (the masks depend on W):
//           [x hi (8 bf16) | x mid (8 bf16) | x lo / 2 (8 bf16) | 16 mask bytes]; bit q = 4 ch + t of byte n = m^ch[16 t + n].
//           Row n_rows is all zero: what idle slots fetch.
//   waves   persistent, over the handle's item streams (acm_csr_build_item_streams): a wave reads its parameters once, walks one
//           linear id stream through its quads of four items and fetches ahead across item and quad boundaries.
//   batch   32 neighbours of ONE work item per wave step: two 16-byte fetches per lane (a neighbour's row = four lanes) ->
//           2 KB of wave-private LDS -> operands.  Lane (g = lane >> 4, m = lane & 15), contraction slot e = 0..7 <-> the
//           neighbour in LDS row 4 e + g (any bijection serves, both operands use this one: bank-conflict free):
//             A  (16 x 32, x):     rows 0..7 = hi of the features, rows 8..15 = mid;   A2: rows 0..7 = rows 8..15 = lo / 2
//             B  (32 x 16, masks): column n = m of tile q: ((byte pair) & (0x00010001 << q)) * (0x3F80 >> q) = two bf16 0 / 1
//           16 MFMAs (8 tiles x {A, A2}, both into the same accumulator); D[4 g + r][n]: lane (g, n) holds
//           V[c = 16 t + n][f = 4 (g & 1) + r], hi + lo / 2 in lane rows 0, 1 and mid + lo / 2 in rows 2, 3.
//   item end forward:  S[c] = sum over the four lane rows of sum_r W[4 (g & 1) + r][c] * D[r]; the wave's four items
//           leave their sums in the four lane rows and share the epilogue of acm_conv_acmii.hip (head, mix, post-op).
//   item end backward: acc += (+-rs_i G[i, c]) * D -- 32 accumulators per lane, splits and waves summed once per workgroup.
#include "acm_conv_device.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef u32x4 u32x4_ma __attribute__((may_alias));
typedef unsigned short u16_ma __attribute__((may_alias));
typedef unsigned char u8_ma __attribute__((may_alias));

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bitsf(unsigned x) { return __builtin_bit_cast(float, x); }
// two fp32 -> their upper halves as one dword (the first in the low half)
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ------------------------------------------------------------------ the table
// Sixteen lanes per node: lane n computes z[ch][16 t + n] (its mask byte) and writes dword n of the row.
__global__ __launch_bounds__(256) void acmii_table_kernel(long n_rows, int f_in, const float* __restrict__ x, long ldx,
                                                          const float* __restrict__ w_low, const float* __restrict__ w_high,
                                                          long ldw, unsigned* __restrict__ table) {
    const int n = threadIdx.x & 15, part = n >> 2, pr = n & 3;
    float w[8][8];                                   // [q = 4 ch + t][f]
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float wv = (q < 4 ? w_low : w_high)[(long)(f < f_in ? f : 0) * ldw + 16 * (q & 3) + n];
            w[q][f] = f < f_in ? wv : 0.f;
        }
    const long stride = (long)gridDim.x * 16;
    for (long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4); row <= n_rows; row += stride) {
        unsigned dw = 0;
        if (row < n_rows) {                          // uniform over the sixteen lanes of a node
            float xv[8];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float2 v = *reinterpret_cast<const float2*>(x + row * ldx + 2 * h);
                xv[2 * h] = v.x, xv[2 * h + 1] = v.y;
            }
            unsigned b = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float z = 0.f;
#pragma unroll
                for (int f = 0; f < 8; ++f) z = fmaf(xv[f], w[q][f], z);
                b |= (z > 0.f ? 1u : 0u) << q;
            }
            const float a = pr == 0 ? xv[0] : (pr == 1 ? xv[2] : (pr == 2 ? xv[4] : xv[6]));
            const float c = pr == 0 ? xv[1] : (pr == 1 ? xv[3] : (pr == 2 ? xv[5] : xv[7]));
            const float ra = a - bitsf(fbits(a) & 0xFFFF0000u), rc = c - bitsf(fbits(c) & 0xFFFF0000u);
            const float sa = ra - bitsf(fbits(ra) & 0xFFFF0000u), sc = rc - bitsf(fbits(rc) & 0xFFFF0000u);
            // (the third part is stored HALVED -- exact, a power of two -- because the kernels add it twice: once beside hi,
            //  once beside mid, see v_batch)
            const unsigned hi = pack_hi16(fbits(a), fbits(c)), mid = pack_hi16(fbits(ra), fbits(rc)),
                           lo = pack_hi16(fbits(0.5f * sa), fbits(0.5f * sc));
            const unsigned m0 = __shfl(b, 4 * pr, 16), m1 = __shfl(b, 4 * pr + 1, 16), m2 = __shfl(b, 4 * pr + 2, 16),
                           m3 = __shfl(b, 4 * pr + 3, 16);
            const unsigned md = m0 | (m1 << 8) | (m2 << 16) | (m3 << 24);
            dw = part == 0 ? hi : (part == 1 ? mid : (part == 2 ? lo : md));
        }
        table[row * 16 + n] = dw;
    }
}

// ------------------------------------------------------------------ the wave's batch sequence
// Persistent waves over the handle's item streams (acm_csr_build_item_streams, acm_common.h: AcmItemStreams): a wave walks
// ONE linear id stream -- batch b = ids[32 b .. 32 b + 31], idle slots hold the zero row's index -- through its quads of
// four items; it reads its parameters once and fetches ahead across item and quad boundaries.
struct VStreamView {
    const int32_t* ids;
    const int32_t* quads;
    const int32_t* waves;
    int n_waves;
};

__device__ __forceinline__ int sel4(int a0, int a1, int a2, int a3, int i) { return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3)); }
// 16 bytes of table row j: a 32-bit byte offset beside the scalar base (the launch checks that the table is below 4 GB)
__device__ __forceinline__ u32x4 v_row(const u32x4* __restrict__ table, int j, int lane) {
    const unsigned off = (unsigned)j * 64u + (unsigned)(lane & 3) * 16u;
    return *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(table) + off);
}

// One batch: rows -> LDS -> operands -> 16 MFMAs.  `lds` = the wave's 2 KB; only this wave touches it and its LDS
// instructions execute in order, so no barrier: the accesses alias (may_alias types) and the compiler keeps their order.
// Lane (g, m), contraction slot e <-> the neighbour in LDS row 4 e + g (both operands: any bijection serves; this one is
// bank-conflict free).  A = [hi (rows 0..7) | mid (rows 8..15)], A2 = [lo / 2 | lo / 2] (the table stores the halved third
// part, so no lane has to blank its rows); both products go to the SAME accumulator (exact products, one fp32 sum): D row
// m < 8 = sum mask (hi + lo / 2) of feature m, row m >= 8 = sum mask (mid + lo / 2) of feature m - 8; every consumer adds the two.
// FIRST: the item's first batch -- the accumulators start from the MFMA's zero C operand instead of 32 register moves.
template <bool FIRST>
__device__ __forceinline__ void v_batch(unsigned char* lds, int lane, u32x4 r0, u32x4 r1, f32x4 (&d)[8]) {
    u32x4_ma* l4 = reinterpret_cast<u32x4_ma*>(lds);
    l4[lane] = r0;
    l4[64 + lane] = r1;
    const int g = lane >> 4, m = lane & 15;
    const u16_ma* l16 = reinterpret_cast<const u16_ma*>(lds);
    const u8_ma* l8 = reinterpret_cast<const u8_ma*>(lds);
    u32x4 A, A2;
    unsigned wp[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int ra = 4 * (2 * p) + g, rb = 4 * (2 * p + 1) + g;
        A[p] = (unsigned)l16[ra * 32 + m] | ((unsigned)l16[rb * 32 + m] << 16);
        A2[p] = (unsigned)l16[ra * 32 + 16 + (m & 7)] | ((unsigned)l16[rb * 32 + 16 + (m & 7)] << 16);
        wp[p] = (unsigned)l8[ra * 64 + 48 + m] | ((unsigned)l8[rb * 64 + 48 + m] << 16);
    }
    u32x4 B[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int p = 0; p < 4; ++p) B[q][p] = __umul24(wp[p] & (0x00010001u << q), 0x3F80u >> q);
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        d[q] = mma(A, B[q], FIRST ? zero4 : d[q]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) d[q] = mma(A2, B[q], d[q]);
}

// The memory pipeline of a wave in STATIC registers: table rows V_RD batches ahead, column ids V_RD + V_JD batches ahead of
// the batch in hand.  The loop over a quad's batches is unrolled over the ring (no value moves between registers, a load is
// awaited only where it is used); when a quad ends in the middle of the ring the live entries are rotated to phase 0 -- a
// handful of moves per four rows -- so that the code after the loop (the rows' epilogue) exists once.
#ifndef ACM_V_RD
#define ACM_V_RD 2
#endif
#ifndef ACM_V_JD
#define ACM_V_JD ACM_V_RD
#endif
constexpr int V_RD = ACM_V_RD, V_JD = ACM_V_JD, V_RING = V_RD + V_JD;

// The wave's quads [qb, qe) with their batches.  `quad_begin(id)` (id = lane row kq's item {row, slot, batches, flags}) runs before
// a quad's first batch, `item_begin(u)` before item u's first batch, `item_end(u, d)` after its last one (also for items
// without batches), `quad_end(id)` after the quad's last item.
template <class QuadBegin, class ItemBegin, class ItemEnd, class QuadEnd>
__device__ __forceinline__ void v_wave_quads(const int32_t* __restrict__ ids, const int32_t* __restrict__ quads, int qb, int qe,
                                             int first_batch, const u32x4* __restrict__ table, unsigned char* lds, int lane,
                                             QuadBegin&& quad_begin, ItemBegin&& item_begin, ItemEnd&& item_end, QuadEnd&& quad_end) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int kq = lane >> 4;
    int J0[V_RING], J1[V_RING];
    u32x4 R0[V_RD], R1[V_RD];
    unsigned ido = ((unsigned)first_batch * 32u + (unsigned)(lane >> 2)) * 4u;   // byte offset of the batch whose ids are fetched next
    auto id_at = [&](unsigned o) { return *reinterpret_cast<const int32_t*>(reinterpret_cast<const unsigned char*>(ids) + o); };
#pragma unroll
    for (int s = 0; s < V_RING; ++s) J0[s] = id_at(ido), J1[s] = id_at(ido + 64u), ido += 128u;
#pragma unroll
    for (int s = 0; s < V_RD; ++s) R0[s] = v_row(table, J0[s], lane), R1[s] = v_row(table, J1[s], lane);
    int4 dn = *reinterpret_cast<const int4*>(quads + 16 * (long)qb + 4 * kq);
    for (int qd = qb; qd < qe; ++qd) {
        const int4 id = dn;                                  // {row, slot, batches, flags} of lane row kq's item
        dn = *reinterpret_cast<const int4*>(quads + 16 * (long)(qd + 1 < qe ? qd + 1 : qd) + 4 * kq);    // a quad ahead
        const int nb0 = __builtin_amdgcn_readlane(id.z, 0), nb1 = __builtin_amdgcn_readlane(id.z, 16),
                  nb2 = __builtin_amdgcn_readlane(id.z, 32), nb3 = __builtin_amdgcn_readlane(id.z, 48);
        quad_begin(id);
        f32x4 d[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] = zero4;
        int u = 0, left = nb0;
        item_begin(0);
        bool more = true, fresh = true;               // fresh: the item has not run a batch yet
        while (more) {
#pragma unroll
            for (int s = 0; s < V_RING; ++s) {
                if (more) {
                    while (left == 0) {
                        if (fresh) {                   // an item without neighbours: its sums are zero
#pragma unroll
                            for (int q = 0; q < 8; ++q) d[q] = zero4;
                        }
                        item_end(u, d);
                        if (++u == 4) {
                            more = false;
                            break;
                        }
                        left = sel4(nb0, nb1, nb2, nb3, u);
                        item_begin(u);
                        fresh = true;
                    }
                    if (more) {
                        const u32x4 a0 = R0[s % V_RD], a1 = R1[s % V_RD];
                        R0[s % V_RD] = v_row(table, J0[(s + V_RD) % V_RING], lane);       // the batch V_RD ahead
                        R1[s % V_RD] = v_row(table, J1[(s + V_RD) % V_RING], lane);
                        J0[s] = id_at(ido), J1[s] = id_at(ido + 64u), ido += 128u;        // the ids V_RING ahead
                        if (fresh) v_batch<true>(lds, lane, a0, a1, d);
                        else v_batch<false>(lds, lane, a0, a1, d);
                        fresh = false;
                        --left;
                    } else if (s != 0) {                   // the quad ended at phase s: rotate the live entries to phase 0
                        u32x4 t0[V_RD], t1[V_RD];
                        int tj0[V_JD], tj1[V_JD];
#pragma unroll
                        for (int i = 0; i < V_RD; ++i) t0[i] = R0[(s + i) % V_RD], t1[i] = R1[(s + i) % V_RD];
#pragma unroll
                        for (int i = 0; i < V_JD; ++i) tj0[i] = J0[(s + V_RD + i) % V_RING], tj1[i] = J1[(s + V_RD + i) % V_RING];
#pragma unroll
                        for (int i = 0; i < V_RD; ++i) R0[i] = t0[i], R1[i] = t1[i];
#pragma unroll
                        for (int i = 0; i < V_JD; ++i) J0[V_RD + i] = tj0[i], J1[V_RD + i] = tj1[i];
                    }
                }
            }
        }
        quad_end(id);
    }
}

// the wave's range of quads and its first batch (wave-uniform)
__device__ __forceinline__ void v_wave(const VStreamView& sv, int w, int& qb, int& qe, int& first_batch) {
    const int4 wd = *reinterpret_cast<const int4*>(sv.waves + 4 * (long)w);
    qb = __builtin_amdgcn_readfirstlane(wd.x);
    qe = __builtin_amdgcn_readfirstlane(wd.y);
    first_batch = __builtin_amdgcn_readfirstlane(wd.z);
}
// ------------------------------------------------------------------ forward
template <int K>
__global__ __launch_bounds__(256) void acmii_v_fwd_kernel(acm_conv_acmii_fwd_t p, VStreamView sv, const AcmLongRow* __restrict__ long_rows,
                                                          const u32x4* __restrict__ table, float* __restrict__ partial) {
    constexpr int T = 8;
    __shared__ __attribute__((aligned(16))) float hlds[3 * K * 64];
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][2048];
    __shared__ float spl[4][4 * T * 64];
    const int F = 64;
    stage_head_params<K>(hlds, p.att_vec, p.ln_weight, p.ln_bias, p.layernorm, F);
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    const int ga = i >> 2, ra = i & 3;           // the self term's A-operand row: item ga of the quad (acm_conv_acmii.hip)
    // contraction weights of this lane's four D rows: W_ch[4 (kq & 1) + r][16 t + i]
    float wc[T][4];
#pragma unroll
    for (int q = 0; q < T; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 4 * (kq & 1) + r;
            const float wv = (q < 4 ? p.w_low : p.w_high)[(long)(f < p.f_in ? f : 0) * p.ld_w + 16 * (q & 3) + i];   // branch-free
            wc[q][r] = f < p.f_in ? wv : 0.f;
        }
    // B operands of the rows' own projections relu(x_i [W_H | W_I]) on the fp32 matrix pipe: feature f = 2 kq + s
    float bs[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int f = 2 * kq + s;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float wv = (t < 4 ? p.w_high : p.w_mlp)[(long)(f < p.f_in ? f : 0) * p.ld_w + 16 * (t & 3) + i];
            bs[s][t] = f < p.f_in ? wv : 0.f;
        }
    }
    float mixm[K * K];
#pragma unroll
    for (int q = 0; q < K * K; ++q) mixm[q] = p.att_mix[q];
    const AcmDropCtx dc = acm_drop_ctx(p.post_drop);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= sv.n_waves) return;
    int qb, qe, first_batch;
    v_wave(sv, w, qb, qe, first_batch);
    if (qb >= qe) return;
    // item u's contraction per lane row goes to the wave's LDS slab [u][t][lane]; at the quad's end lane (kq, i) adds the four
    // lane rows of ITS item kq -- 4 reads + 3 adds per tile instead of a four-row swap-and-add reduction per item and tile
    float* spw = spl[threadIdx.x >> 6];
    v_wave_quads(
        sv.ids, sv.quads, qb, qe, first_batch, table, stage[threadIdx.x >> 6], lane, [](const int4&) {}, [](int) {},
        [&](int u, const f32x4 (&d)[8]) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float s = d[t][0] * wc[t][0];
#pragma unroll
                for (int r = 1; r < 4; ++r) s = fmaf(d[t][r], wc[t][r], s);
                spw[(u * T + t) * 64 + lane] = s;
            }
        },
        [&](const int4& id) {
            float acc[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float* q = spw + (kq * T + t) * 64 + i;
                acc[t] = (q[0] + q[16]) + (q[32] + q[48]);
            }
            const bool valid_d = (id.w & 1) != 0;
            const int row = id.x, slot = id.y;
            const long rr = valid_d ? row : 0;
            // the rows' own projected features relu(x_i [W_H | W_I]): A row 4 g of the operand carries item g's input row
            float zs[8];
            {
                const int row_a = __shfl(row, 16 * ga), flags_a = __shfl(id.w, 16 * ga);
                const float2 xi = ((flags_a & 1) && ra == 0) ? *reinterpret_cast<const float2*>(p.xs + (long)row_a * p.ld_xs + 2 * kq)
                                                             : make_float2(0.f, 0.f);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(xi.x, bs[0][t], zero4, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(xi.y, bs[1][t], d, 0, 0, 0);
                    zs[t] = fmaxf(d[0], 0.f);
                }
            }
            // ---- per 16-lane row: its item (row, slot); from here on as acm_conv_acmii.hip
            const bool owner = (id.w & 2) != 0;                // a whole row, or the first piece of a long one
            if (owner) {
                if (p.zlh) {                               // the H half only (self term of a long row's fix-up); the L half is not computed here
#pragma unroll
                    for (int t = 0; t < 4; ++t) p.zlh[rr * p.ld_zlh + F + 16 * t + i] = zs[t];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) p.zi[rr * p.ld_zi + 16 * t + i] = zs[4 + t];
            }
            if (valid_d && slot >= 0) {                        // a piece of a long row: raw sums to its slot
                float* ps = partial + (long)slot * (2 * F);
#pragma unroll
                for (int t = 0; t < T; ++t) ps[16 * t + i] = acc[t];
            }
            const bool full = valid_d && slot < 0;
            const float rs = p.row_scale[rr];
            float H[K][4], pre[3][4];
            const float dg = K == 4 ? p.deg[rr] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                pre[0][t] = rs * acc[t];
                pre[1][t] = zs[t] - rs * acc[4 + t];
                H[0][t] = pre[0][t];                           // ACMII: no ReLU after the filter
                H[1][t] = pre[1][t];
                H[2][t] = zs[4 + t];
                if (K == 4) {                                  // structure channel: relu(A S) = relu(deg (A_low S) - S), ps = A_low S
                    pre[2][t] = dg * p.ps[rr * p.ld_ps + i + 16 * t] - p.ss[rr * p.ld_ss + i + 16 * t];
                    H[K - 1][t] = fmaxf(pre[2][t], 0.f);
                }
            }
            RowHead<K> rh;
            row_head<K>(hlds, mixm, acm_opaque(i), F, p.layernorm != 0, H, rh);
            float df[4];
            acm_drop4(dc, rr, i, df);
            if (full) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int col = i + 16 * t;
                    float o = rh.alpha[0] * H[0][t] + rh.alpha[1] * H[1][t] + rh.alpha[2] * H[2][t];
                    if (K == 4) o = fmaf(rh.alpha[K - 1], H[K - 1][t], o);
                    o *= p.scale;
                    if (p.post_relu) o = fmaxf(o, 0.f);
                    if (p.post_scale) o *= p.post_scale[rr * p.ld_post_scale + col];
                    if (p.post_drop.p > 0.f) o *= df[t];
                    p.out[rr * p.ld_out + col] = o;
                    p.pre[rr * p.ld_pre + col] = pre[0][t];
                    p.pre[rr * p.ld_pre + F + col] = pre[1][t];
                    if (K == 4) p.pre[rr * p.ld_pre + 2 * F + col] = pre[2][t];
                }
                if (i == 0)
                    *reinterpret_cast<float4*>(p.att + rr * 4) =
                        make_float4(rh.alpha[0], rh.alpha[1], rh.alpha[2], K == 4 ? rh.alpha[K - 1] : 0.f);
            }
        });
}

// ------------------------------------------------------------------ backward: dW_L, dW_H, dW_I
// partial: [48 groups][n_blocks][32]: group = column / 32 of the 1536 columns (ch, f, c) = 512 ch + 64 f + c, ch = L, H, I.
// The row-local terms -- the high-pass channel's self term G_H[i, c] m^H_i[c] x_i[f] and the identity channel's
// dZ_I[i, c] x_i[f] -- are plain fp32 FMAs: lane (kq, i) owns features 2 kq, 2 kq + 1 of column i of every tile.
__global__ __launch_bounds__(256) void acmii_v_bwd_kernel(acm_conv_acmii_bwd_t p, VStreamView sv, const u32x4* __restrict__ table,
                                                          float* __restrict__ partial) {
    constexpr int T = 8;
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][2048];
    __shared__ float red[4][1536];
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4, wv = threadIdx.x >> 6;
    const int w = blockIdx.x * 4 + wv;
    int qb = 0, qe = 0, first_batch = 0;
    if (w < sv.n_waves) v_wave(sv, w, qb, qe, first_batch);
    float accd[T][4], accs[4][2], acci[4][2];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) accd[t][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) accs[t][0] = accs[t][1] = acci[t][0] = acci[t][1] = 0.f;
    if (qb < qe) {
        // all four lane rows work on the same item: lane (kq, i) owns the D rows 4 kq + r (hi + lo: kq 0, 1; mid: kq 2, 3;
        // features 4 (kq & 1) + r), column i of every tile
        int rw0 = 0, rw1 = 0, rw2 = 0, rw3 = 0, fl0 = 0, fl1 = 0, fl2 = 0, fl3 = 0;
        float gu[T], gi[4], rsu = 0.f, own = 0.f;
        float2 xv = make_float2(0.f, 0.f);
        unsigned mself = 0;
        v_wave_quads(
            sv.ids, sv.quads, qb, qe, first_batch, table, stage[wv], lane,
            [&](const int4& id) {
                rw0 = __builtin_amdgcn_readlane(id.x, 0), rw1 = __builtin_amdgcn_readlane(id.x, 16);
                rw2 = __builtin_amdgcn_readlane(id.x, 32), rw3 = __builtin_amdgcn_readlane(id.x, 48);
                fl0 = __builtin_amdgcn_readlane(id.w, 0), fl1 = __builtin_amdgcn_readlane(id.w, 16);
                fl2 = __builtin_amdgcn_readlane(id.w, 32), fl3 = __builtin_amdgcn_readlane(id.w, 48);
            },
            [&](int u) {             // the item's G rows, scale, own input and masks: requested before its batches
                const long ru = sel4(rw0, rw1, rw2, rw3, u);     // (row 0 for an absent item: scaled by zero)
                const int fl = sel4(fl0, fl1, fl2, fl3, u);
                rsu = (fl & 1) ? p.row_scale[ru] : 0.f;
                own = (fl & 2) ? 1.f : 0.f;
#pragma unroll
                for (int t = 0; t < T; ++t)
                    gu[t] = t < 4 ? p.g_low[ru * p.ld_g_low + 16 * t + i] : p.g_high[ru * p.ld_g_high + 16 * (t & 3) + i];
#pragma unroll
                for (int t = 0; t < 4; ++t) gi[t] = p.g_mlp[ru * p.ld_g_mlp + 16 * t + i];
                xv = *reinterpret_cast<const float2*>(p.x + ru * p.ld_x + 2 * kq);
                mself = reinterpret_cast<const unsigned char*>(table)[(ru + p.self_offset) * 64 + 48 + i];
            },
            [&](int u, const f32x4 (&d)[8]) {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float cf = (t < 4 ? rsu : -rsu) * gu[t];
#pragma unroll
                    for (int r = 0; r < 4; ++r) accd[t][r] = fmaf(cf, d[t][r], accd[t][r]);
                }
                const float x0 = own * xv.x, x1 = own * xv.y;      // pieces that do not own the row: nothing
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float gm = ((mself >> (4 + t)) & 1u) ? gu[4 + t] : 0.f;
                    accs[t][0] = fmaf(gm, x0, accs[t][0]);
                    accs[t][1] = fmaf(gm, x1, accs[t][1]);
                    acci[t][0] = fmaf(gi[t], x0, acci[t][0]);
                    acci[t][1] = fmaf(gi[t], x1, acci[t][1]);
                }
            },
            [](const int4&) {});
    }
    // splits: hi + lo (lane rows 0, 1) + mid (rows 2, 3) -> lane rows 0, 1 hold features 4 kq + r
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const acm_u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(accd[t][r]), __float_as_uint(accd[t][r]), false, false);
            const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            if (kq < 2) red[wv][512 * (t >> 2) + 64 * (4 * kq + r) + 16 * (t & 3) + i] = tot;
        }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            red[wv][512 + 64 * (2 * kq + s) + 16 * t + i] += accs[t][s];
            red[wv][1024 + 64 * (2 * kq + s) + 16 * t + i] = acci[t][s];
        }
    __syncthreads();
    const long gstride = (long)gridDim.x * 32;
#pragma unroll
    for (int h = 0; h < 6; ++h) {
        const int col = threadIdx.x + 256 * h;
        const float s = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        partial[(long)(col >> 5) * gstride + (long)blockIdx.x * 32 + (col & 31)] = s;
    }
}

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" int acm_acmii_table_bytes(int64_t n_rows, size_t* bytes) {
    ACM_REQUIRE(bytes && n_rows >= 0, ACM_EINVAL, "acm_acmii_table_bytes: NULL argument / negative row count");
    *bytes = (size_t)(n_rows + 1) * 64;
    return ACM_OK;
}

extern "C" int acm_acmii_table(int64_t n_rows, int f_in, const float* x, int64_t ld_x, const float* w_low, const float* w_high,
                               int64_t ld_w, void* table, size_t table_bytes, acm_stream_t stream) {
    ACM_REQUIRE(x && w_low && w_high && table, ACM_EINVAL, "acm_acmii_table: NULL argument");
    ACM_REQUIRE(f_in >= 1 && f_in <= 8 && ld_x >= 8 && ld_x % 2 == 0 && ((uintptr_t)x) % 8 == 0 && ld_w >= 64, ACM_EUNSUPPORTED,
                "acm_acmii_table: f_in %d (needs f_in <= 8, rows of 8 zero-padded floats, 8-byte aligned; 64 output columns)", f_in);
    ACM_REQUIRE(table_bytes >= (size_t)(n_rows + 1) * 64 && ((uintptr_t)table) % 16 == 0, ACM_ENOMEM,
                "acm_acmii_table: table of %zu B < %zu B, or not 16-byte aligned", table_bytes, (size_t)(n_rows + 1) * 64);
    ACM_REQUIRE(n_rows < (int64_t)1 << 31, ACM_EUNSUPPORTED, "acm_acmii_table: %lld rows", (long long)n_rows);
    const long groups = n_rows + 1;
    const int grid = (int)((groups + 15) / 16 < 4096 ? (groups + 15) / 16 : 4096);
    hipLaunchKernelGGL(acmii_table_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (long)n_rows, f_in, x, (long)ld_x, w_low,
                       w_high, (long)ld_w, (unsigned*)table);
    ACM_CHECK_HIP(hipGetLastError());
    return ACM_OK;
}

static int acmii_v_check_operator(const acm_csr_t* a, const void* table, const char* who) {
    ACM_REQUIRE(a && table, ACM_EINVAL, "%s: NULL argument", who);
    ACM_REQUIRE(a->vals == nullptr, ACM_EUNSUPPORTED, "%s: pattern-only operators only (explicit values scale the inputs, not the masks)", who);
    ACM_REQUIRE(a->nnz > 0 && (a->n_cols + 1) * 64 < ((int64_t)1 << 32), ACM_EUNSUPPORTED, "%s: empty operator / a table of 4 GB or more", who);
    ACM_REQUIRE(((uintptr_t)table) % 16 == 0, ACM_EINVAL, "%s: table not 16-byte aligned", who);
    ACM_REQUIRE(a->n_long == 0 || a->long_index, ACM_EUNSUPPORTED, "%s: handle without a long-row index", who);
    ACM_REQUIRE(a->item_streams, ACM_EINVAL, "%s: the operator has no item streams (acm_csr_build_item_streams)", who);
    return ACM_OK;
}

static VStreamView stream_view(const AcmItemStreams* t) {
    VStreamView v;
    v.ids = t->ids, v.quads = t->quads, v.waves = t->waves, v.n_waves = t->n_waves;
    return v;
}

// The forward on a table acm_acmii_table has just written for the SAME x and weights.  p as for acm_conv_acmii_fwd; p->xg is not
// read (the table replaces it), p->zlh may be NULL (only its high-pass half would be written).  Long rows: the pieces' partial
// sums are combined by acm_conv_acmii_fwd's fix-up launch, which reads zlh's high-pass half -- so zlh is required with long rows.
extern "C" int acm_conv_acmii_v_fwd(const acm_csr_t* a, const acm_conv_acmii_fwd_t* p, const void* table, void* workspace,
                                    size_t workspace_bytes, acm_stream_t stream);
int acm_acmii_fixup_launch(const acm_csr_t* a, const acm_conv_acmii_fwd_t* p, const float* partial, hipStream_t s);   // acm_conv_acmii.hip

extern "C" int acm_conv_acmii_v_fwd(const acm_csr_t* a, const acm_conv_acmii_fwd_t* p, const void* table, void* workspace,
                                    size_t workspace_bytes, acm_stream_t stream) {
    ACM_REQUIRE(p, ACM_EINVAL, "acm_conv_acmii_v_fwd: NULL argument");
    const int st = acmii_v_check_operator(a, table, "acm_conv_acmii_v_fwd");
    if (st != ACM_OK) return st;
    ACM_REQUIRE(p->f_out == 64 && p->f_in >= 1 && p->f_in <= 8 && p->f_pad == 8, ACM_EUNSUPPORTED,
                "acm_conv_acmii_v_fwd: f_in %d f_pad %d f_out %d (needs f_in <= 8 = f_pad, f_out = 64)", p->f_in, p->f_pad, p->f_out);
    ACM_REQUIRE(p->xs && p->w_low && p->w_high && p->w_mlp && p->att_mix && p->out && p->pre && p->att && p->zi && p->row_scale,
                ACM_EINVAL, "acm_conv_acmii_v_fwd: NULL tensor pointer (row_scale is required: pattern-only operator)");
    ACM_REQUIRE(((uintptr_t)p->xs) % 8 == 0 && p->ld_xs % 2 == 0 && p->ld_xs >= p->f_pad && ((uintptr_t)p->att) % 16 == 0, ACM_EINVAL,
                "acm_conv_acmii_v_fwd: xs rows must be 8-byte aligned and f_pad long, att 16-byte aligned");
    const int K = p->n_channels;
    ACM_REQUIRE(K == 3 || K == 4, ACM_ESHAPE, "acm_conv_acmii_v_fwd: n_channels %d", K);
    ACM_REQUIRE(p->ld_w >= 64 && p->ld_out >= 64 && p->ld_pre >= 64 * (K - 1) && p->ld_zi >= 64 && (!p->zlh || p->ld_zlh >= 128), ACM_ESHAPE,
                "acm_conv_acmii_v_fwd: leading dimension too small");
    ACM_REQUIRE(K == 3 || (p->ps && p->ss && p->deg && p->ld_ps >= 64 && p->ld_ss >= 64), ACM_EINVAL,
                "acm_conv_acmii_v_fwd: structure-channel pointers NULL / leading dimensions too small");
    ACM_REQUIRE(a->n_long == 0 || p->zlh, ACM_EINVAL, "acm_conv_acmii_v_fwd: zlh is required when the operator has long rows");
    for (int c = 0; c < K; ++c) {
        ACM_REQUIRE(p->att_vec[c], ACM_EINVAL, "acm_conv_acmii_v_fwd: att_vec[%d] NULL", c);
        ACM_REQUIRE(!p->layernorm || (p->ln_weight[c] && p->ln_bias[c]), ACM_EINVAL, "acm_conv_acmii_v_fwd: LayerNorm pointers NULL");
    }
    size_t need = 0;
    acm_conv_acmii_fwd_workspace_bytes(a, &need);
    ACM_REQUIRE(workspace && workspace_bytes >= need, ACM_ENOMEM, "acm_conv_acmii_v_fwd: workspace %zu B < required %zu B", workspace_bytes, need);
    if (a->n_rows == 0 || a->n_items == 0) return ACM_OK;
    hipStream_t s = (hipStream_t)stream;
    const VStreamView sv = stream_view(a->item_streams);
    const int grid = a->item_streams->n_waves / 4;           // persistent waves: the streams were cut for exactly these
    if (K == 3) hipLaunchKernelGGL(acmii_v_fwd_kernel<3>, dim3(grid), dim3(256), 0, s, *p, sv, a->long_rows, (const u32x4*)table, (float*)workspace);
    else hipLaunchKernelGGL(acmii_v_fwd_kernel<4>, dim3(grid), dim3(256), 0, s, *p, sv, a->long_rows, (const u32x4*)table, (float*)workspace);
    ACM_CHECK_HIP(hipGetLastError());
    if (a->n_long) return acm_acmii_fixup_launch(a, p, (const float*)workspace, s);
    return ACM_OK;
}

extern "C" int acm_conv_acmii_v_bwd_workspace_bytes(const acm_csr_t* a, size_t* bytes) {
    ACM_REQUIRE(a && bytes, ACM_EINVAL, "acm_conv_acmii_v_bwd_workspace_bytes: NULL argument");
    ACM_REQUIRE(a->item_streams, ACM_EINVAL, "acm_conv_acmii_v_bwd_workspace_bytes: the operator has no item streams (acm_csr_build_item_streams)");
    *bytes = (size_t)(a->item_streams->n_waves / 4) * 1536 * sizeof(float);       // one slab of 1536 sums per workgroup
    return ACM_OK;
}

extern "C" int acm_conv_acmii_v_bwd(const acm_csr_t* a, const acm_conv_acmii_bwd_t* p, void* workspace, size_t workspace_bytes,
                                    acm_stream_t stream) {
    ACM_REQUIRE(p, ACM_EINVAL, "acm_conv_acmii_v_bwd: NULL argument");
    const int st = acmii_v_check_operator(a, p->table, "acm_conv_acmii_v_bwd");
    if (st != ACM_OK) return st;
    ACM_REQUIRE(p->f_in >= 1 && p->f_in <= 8, ACM_EUNSUPPORTED, "acm_conv_acmii_v_bwd: f_in %d", p->f_in);
    ACM_REQUIRE(p->self_offset >= 0 && p->self_offset + a->n_rows <= a->n_cols, ACM_EINVAL,
                "acm_conv_acmii_v_bwd: self_offset %lld + %lld rows exceed the table's %lld rows", (long long)p->self_offset,
                (long long)a->n_rows, (long long)a->n_cols);
    ACM_REQUIRE(p->g_low && p->g_high && p->g_mlp && p->x && p->row_scale && p->d_w_low && p->d_w_high && p->d_w_mlp, ACM_EINVAL,
                "acm_conv_acmii_v_bwd: NULL tensor pointer");
    ACM_REQUIRE(p->ld_g_low >= 64 && p->ld_g_high >= 64 && p->ld_g_mlp >= 64 && p->ld_dw >= 64 && p->ld_x >= 8 && p->ld_x % 2 == 0 &&
                    ((uintptr_t)p->x) % 8 == 0, ACM_ESHAPE, "acm_conv_acmii_v_bwd: leading dimension too small / x rows not 8-byte aligned");
    size_t need = 0;
    const int stw = acm_conv_acmii_v_bwd_workspace_bytes(a, &need);
    if (stw != ACM_OK) return stw;
    ACM_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace) % 16 == 0, ACM_ENOMEM,
                "acm_conv_acmii_v_bwd: workspace %zu B < required %zu B (or not 16-byte aligned)", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    if (a->n_rows == 0 || a->n_items == 0) {
        ACM_CHECK_HIP(hipMemset2DAsync(p->d_w_low, (size_t)p->ld_dw * 4, 0, 64 * 4, (size_t)p->f_in, s));
        ACM_CHECK_HIP(hipMemset2DAsync(p->d_w_high, (size_t)p->ld_dw * 4, 0, 64 * 4, (size_t)p->f_in, s));
        ACM_CHECK_HIP(hipMemset2DAsync(p->d_w_mlp, (size_t)p->ld_dw * 4, 0, 64 * 4, (size_t)p->f_in, s));
        return ACM_OK;
    }
    const VStreamView sv = stream_view(a->item_streams);
    const int grid = a->item_streams->n_waves / 4;
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(acmii_v_bwd_kernel, dim3(grid), dim3(256), 0, s, *p, sv, (const u32x4*)p->table, partial);
    ACM_CHECK_HIP(hipGetLastError());
    const int len = p->f_in * 64;
    const acm_reduce_seg_t segs[3] = {{partial, grid, 32, 0, len, p->d_w_low, 64, 0, p->ld_dw, 0, grid * 32, 0},
                                      {partial, grid, 32, 512, len, p->d_w_high, 64, 0, p->ld_dw, 0, grid * 32, 0},
                                      {partial, grid, 32, 1024, len, p->d_w_mlp, 64, 0, p->ld_dw, 0, grid * 32, 0}};
    return acm_reduce_emit(p->defer, segs, 3, s);
}
