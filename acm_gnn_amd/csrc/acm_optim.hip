// Fused Adam / AdamW update over a list of parameter tensors (gfx950).
// torch.optim's multi-tensor path issues ~3 launches per parameter for the bias-correction scalars on top of
// the foreach kernels (~80 launches, ~0.37 ms per step for the 26 parameters of the two-layer model, as much
// as the two sparse products of a layer together).  Here the whole update is one launch per 32 tensors: the
// tensor table travels by value in the kernel arguments (so it is capture-safe: nothing is read from host
// memory at replay), each 256-thread block owns 2048 consecutive elements of one tensor, and the step
// counters live on the device (one fp32 scalar per tensor, the layout torch.optim uses when capturable=True).
//
// With acm_adam_config_t.pending the launch also IS the step's deferred reduction flush (acm_reduce_flush): the first
// blocks of the grid are the second-phase blocks of the pending segments; a block whose sums are elements of a
// parameter's gradient stores them and applies the update to those elements at once (the gradients of the step's
// backward kernels are a few thousand numbers: their "Adam blocks" disappear), every other tensor is updated by the
// blocks behind as before.  One launch and one grid drain less per step.
#include <math.h>

#include "acm_reduce_device.h"
#include "acm_adam_device.h"

namespace {

constexpr int PACK = 40;          // tensors per launch (a 2-layer model with the structure channel has 34-36 with gradients)
constexpr int CHUNK = 2048;       // elements per block and round
constexpr int MAX_BLOCKS = 1024;  // per tensor: a large tensor's blocks take several rounds each (every block ends in an atomic on
                                  // ONE arrival counter, ~11 ns apiece: 5 254 blocks of the twitch-sized N x 64 parameter = 58 us)
// rounds of CHUNK elements a block of a tensor of n elements takes, and the blocks of that tensor
__host__ __device__ inline int adam_rounds(long n) {
    const long r = (n + (long)CHUNK * MAX_BLOCKS - 1) / ((long)CHUNK * MAX_BLOCKS);
    return r > 1 ? (int)r : 1;
}
__host__ __device__ inline int adam_blocks(long n) { return (int)((n + (long)CHUNK * adam_rounds(n) - 1) / ((long)CHUNK * adam_rounds(n))); }

struct AdamPack {
    float* p[PACK];
    const float* g[PACK];
    float* m[PACK];
    float* v[PACK];
    float* step[PACK];
    long numel[PACK];
    int first_block[PACK + 1];
    int n;
};

// block `blk` of the update of pack `pk` (blk counts from the pack's first block)
__device__ __forceinline__ void adam_block(const AdamPack& pk, const AdamScalars& hp, int blk, float* sc) {
    int t = 0;
    while (t + 1 < pk.n && blk >= pk.first_block[t + 1]) ++t;      // uniform: <= 31 scalar compares
    if (threadIdx.x == 0) {
        const double k = (double)pk.step[t][0] + 1.0;
        sc[0] = (float)(hp.lr / (1.0 - pow(hp.beta1, k)));
        sc[1] = (float)sqrt(1.0 - pow(hp.beta2, k));
    }
    __syncthreads();
    const float step_size = sc[0], bc2_sqrt = sc[1];
    const AdamFactors f(hp);
    float* __restrict__ p = pk.p[t];
    const float* __restrict__ g = pk.g[t];
    float* __restrict__ m = pk.m[t];
    float* __restrict__ v = pk.v[t];
    const long n = pk.numel[t];
    const int rounds = adam_rounds(n);
    for (int rd = 0; rd < rounds; ++rd) {
    const long base = ((long)(blk - pk.first_block[t]) * rounds + rd) * CHUNK;
    if (base >= n) break;
    const long end = base + CHUNK < n ? base + CHUNK : n;
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && end - base == CHUNK;
    if (vec) {
#pragma unroll
        for (int r = 0; r < CHUNK / 1024; ++r) {
            const long i = base + r * 1024 + threadIdx.x * 4;
            float4 pp = *reinterpret_cast<const float4*>(p + i), gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<const float4*>(m + i), vv = *reinterpret_cast<const float4*>(v + i);
            adam_one(pp.x, gg.x, mm.x, vv.x, f.decay_eff, f.wd, f.decoupled, f.w1, f.b2, f.w2, step_size, bc2_sqrt, f.eps);
            adam_one(pp.y, gg.y, mm.y, vv.y, f.decay_eff, f.wd, f.decoupled, f.w1, f.b2, f.w2, step_size, bc2_sqrt, f.eps);
            adam_one(pp.z, gg.z, mm.z, vv.z, f.decay_eff, f.wd, f.decoupled, f.w1, f.b2, f.w2, step_size, bc2_sqrt, f.eps);
            adam_one(pp.w, gg.w, mm.w, vv.w, f.decay_eff, f.wd, f.decoupled, f.w1, f.b2, f.w2, step_size, bc2_sqrt, f.eps);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        }
    } else {
        for (long i = base + threadIdx.x; i < end; i += 256) {
            float pp = p[i], mm = m[i], vv = v[i];
            adam_one(pp, g[i], mm, vv, f.decay_eff, f.wd, f.decoupled, f.w1, f.b2, f.w2, step_size, bc2_sqrt, f.eps);
            p[i] = pp;
            m[i] = mm;
            v[i] = vv;
        }
    }
    }
}

// Advance the step counters in the same launch: every block read its counter(s) before it gets here, so the last block
// to arrive may increment them.  Only the arrival counter is shared between blocks (a device-scope atomic; no data is
// handed from block to block, hence no fence); the incremented values are for the NEXT launch, which the kernel
// boundary orders.
__device__ __forceinline__ void adam_arrive(const AdamPack& pk, int* arrive, int64_t* also_advance) {
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int ticket = atomicAdd(arrive, 1);
        s_last = ticket == (int)gridDim.x - 1;
        if (s_last) atomicExch(arrive, 0);
    }
    __syncthreads();
    if (s_last) {
        if ((int)threadIdx.x < pk.n) pk.step[threadIdx.x][0] += 1.0f;
        if (threadIdx.x == 0 && also_advance) also_advance[0] += 1;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(AdamPack pk, AdamScalars hp, int* arrive, int64_t* also_advance) {
    __shared__ float sc[2];
    adam_block(pk, hp, (int)blockIdx.x, sc);
    if (arrive) adam_arrive(pk, arrive, also_advance);
}

// The pending second phases of the step + the update, one grid: blocks [0, pd.blocks) reduce, the rest update the tensors
// no segment writes (pk.first_block counts those only).  `covered` = the tensors whose whole gradient the segments produce.
constexpr int FUSED_SEGS = 28;
struct PendingPack {
    int n, blocks;
    unsigned long long covered;      // bit t: tensor t of the pack (PACK <= 64)
    int first[FUSED_SEGS + 1];
    acm_reduce_seg_t seg[FUSED_SEGS];
};
static_assert(PACK <= 64, "PendingPack::covered is a 64-bit mask");
static_assert(sizeof(AdamPack) + sizeof(PendingPack) + sizeof(AdamScalars) + 16 <= 4096, "kernel arguments: 4 KB");

__global__ __launch_bounds__(256) void adam_flush_kernel(AdamPack pk, PendingPack pd, AdamScalars hp, int* arrive,
                                                         int64_t* also_advance) {
    __shared__ __attribute__((aligned(16))) float red[ACM_REDUCE_LDS];
    if ((int)blockIdx.x < pd.blocks) {
        int e = blockIdx.x, i = 0;
        while (i + 1 < pd.n && e >= pd.first[i + 1]) ++i;        // block-uniform
        const AdamFactors f(hp);
        acm_reduce_block(pd.seg[i], e - pd.first[i], red, [&](float* dst, float gsum) {
            *dst = gsum;
            for (int t = 0; t < pk.n; ++t) {                     // (uniform index: the tables stay in scalar registers)
                if (!((pd.covered >> t) & 1ull)) continue;
                const float* g0 = pk.g[t];
                if (dst < g0 || dst >= g0 + pk.numel[t]) continue;
                const long idx = dst - g0;
                const double k = (double)pk.step[t][0] + 1.0;
                const float step_size = (float)(hp.lr / (1.0 - pow(hp.beta1, k)));
                const float bc2_sqrt = (float)sqrt(1.0 - pow(hp.beta2, k));
                float pp = pk.p[t][idx], mm = pk.m[t][idx], vv = pk.v[t][idx];
                adam_one(pp, gsum, mm, vv, f.decay_eff, f.wd, f.decoupled, f.w1, f.b2, f.w2, step_size, bc2_sqrt, f.eps);
                pk.p[t][idx] = pp, pk.m[t][idx] = mm, pk.v[t][idx] = vv;
                break;
            }
        });
    } else {
        adam_block(pk, hp, (int)blockIdx.x - pd.blocks, red);
    }
    adam_arrive(pk, arrive, also_advance);
}

// after the update of a pack: step_t += 1 for each of its tensors (stream order makes every block of the update
// read the old value)
__global__ void adam_advance_kernel(AdamPack pk, int64_t* also_advance) {
    const int t = threadIdx.x;
    if (t < pk.n) pk.step[t][0] += 1.0f;
    if (t == 0 && also_advance) also_advance[0] += 1;
}

// Elements of segment `sg` whose destination lies in [lo, hi): the destination of element e is
// dst[(e / inner) * outer_stride + blk(e % inner)] -- runs of consecutive addresses of length col_block (or inner).
long seg_overlap(const acm_reduce_seg_t& sg, const float* lo, const float* hi) {
    long hit = 0;
    if (sg.len <= 0) return 0;
    if (sg.outer_stride >= 0 && sg.block_stride >= 0) {
        // most (tensor, segment) pairs are disjoint: compare the segment's bounding range first.  (The run walk below
        // costs two 64-bit divisions per run, and a dW segment with col_block = 1 has one run per element: tens of
        // thousands of iterations per eager step over ~36 tensors x ~10 segments)
        const long jmax = (sg.len - 1) / sg.inner;
        const long maxcol = sg.col_block ? ((long)(sg.inner - 1) / sg.col_block) * sg.block_stride + sg.col_block : (long)sg.inner;
        const float* b0 = sg.dst;
        const float* b1 = sg.dst + jmax * sg.outer_stride + maxcol;
        if (b1 <= lo || b0 >= hi) return 0;
    }
    for (long el = 0; el < sg.len;) {
        const long j = el / sg.inner, q = el % sg.inner;
        long run = sg.col_block ? sg.col_block - q % sg.col_block : sg.inner - q;
        if (run > sg.inner - q) run = sg.inner - q;
        if (run > sg.len - el) run = sg.len - el;
        const long col = sg.col_block ? (q / sg.col_block) * sg.block_stride + q % sg.col_block : q;
        const float* a = sg.dst + j * sg.outer_stride + col;
        const float* b = a + run;
        const float* x = a > lo ? a : lo;
        const float* y = b < hi ? b : hi;
        if (y > x) hit += y - x;
        el += run;
    }
    return hit;
}

// 1: launched (list emptied); 0: the caller flushes and updates separately; < 0: -(error code)
int adam_with_flush(int n_tensors, const acm_adam_tensor_t* tensors, const acm_adam_config_t* cfg, const AdamScalars& hp,
                    hipStream_t s) {
    acm_reduce_list_t* list = cfg->pending;
    if (!cfg->arrive || n_tensors < 1 || n_tensors > PACK) return 0;
    PendingPack pd{};
    for (int i = 0; i < list->n; ++i) {
        const acm_reduce_seg_t& sg = list->segs[i];
        const int ok = acm_reduce_check_segment(sg, i);
        if (ok != ACM_OK) return -ok;
        if (sg.len == 0) continue;
        if (pd.n == FUSED_SEGS) return 0;
        pd.first[pd.n] = pd.blocks;
        pd.seg[pd.n++] = sg;
        pd.blocks += acm_seg_blocks(sg);
    }
    pd.first[pd.n] = pd.blocks;
    if (pd.blocks == 0) return 0;
    AdamPack pk{};
    pk.n = n_tensors;
    int blocks = 0;
    for (int t = 0; t < n_tensors; ++t) {
        const acm_adam_tensor_t& a = tensors[t];
        long hit = 0;
        for (int i = 0; i < pd.n; ++i) hit += seg_overlap(pd.seg[i], a.grad, a.grad + a.numel);
        if (hit != 0 && hit != a.numel) return 0;         // a gradient the segments write in part (or twice): not this way
        pk.p[t] = a.param, pk.g[t] = a.grad, pk.m[t] = a.exp_avg, pk.v[t] = a.exp_avg_sq, pk.step[t] = a.step;
        pk.numel[t] = (long)a.numel;
        pk.first_block[t] = blocks;
        if (hit) pd.covered |= 1ull << t;
        else blocks += adam_blocks((long)a.numel);
    }
    pk.first_block[n_tensors] = blocks;
    // adam_block() finds its tensor by first_block: a covered tensor has no blocks (first_block[t] == first_block[t + 1])
    // and is skipped by the search because the comparison is >=
    hipLaunchKernelGGL(adam_flush_kernel, dim3(pd.blocks + blocks), dim3(256), 0, s, pk, pd, hp, cfg->arrive, cfg->also_advance);
    if (hipGetLastError() != hipSuccess) return -ACM_EHIP;
    list->n = 0;
    return 1;
}

}  // namespace

extern "C" int acm_adam_step(int32_t n_tensors, const acm_adam_tensor_t* tensors, const acm_adam_config_t* cfg,
                             acm_stream_t stream) {
    ACM_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || tensors) && cfg, ACM_EINVAL, "acm_adam_step: NULL argument");
    ACM_REQUIRE(cfg->lr >= 0 && cfg->eps >= 0 && cfg->beta1 >= 0 && cfg->beta1 < 1 && cfg->beta2 >= 0 && cfg->beta2 < 1 &&
                    cfg->weight_decay >= 0,
                ACM_EINVAL, "acm_adam_step: hyper-parameter out of range");
    hipStream_t s = (hipStream_t)stream;
    AdamScalars hp{cfg->lr, cfg->beta1, cfg->beta2, cfg->eps, cfg->weight_decay, cfg->decoupled};
    if (cfg->pending) {
        acm_reduce_list_t* list = cfg->pending;
        ACM_REQUIRE((list->n == 0 || list->segs) && list->n >= 0 && list->n <= list->cap, ACM_EINVAL,
                    "acm_adam_step: inconsistent pending list");
        for (int i = 0; i < n_tensors; ++i)
            ACM_REQUIRE(tensors[i].param && tensors[i].grad && tensors[i].exp_avg && tensors[i].exp_avg_sq && tensors[i].step &&
                            tensors[i].numel >= 0 && tensors[i].numel < ((int64_t)1 << 40), ACM_EINVAL,
                        "acm_adam_step: tensor %d has a NULL pointer or is too large", i);
        int fused = 0;
        if (list->n > 0) fused = adam_with_flush(n_tensors, tensors, cfg, hp, s);
        if (fused < 0) return -fused;
        if (fused) return ACM_OK;
        const int st = acm_reduce_flush(list, stream);           // not one grid after all: flush, then the plain update
        if (st != ACM_OK) return st;
    }
    for (int first = 0; first < n_tensors; first += PACK) {
        AdamPack pk{};
        int blocks = 0;
        pk.n = n_tensors - first < PACK ? n_tensors - first : PACK;
        for (int i = 0; i < pk.n; ++i) {
            const acm_adam_tensor_t& t = tensors[first + i];
            ACM_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq && t.step && t.numel >= 0, ACM_EINVAL,
                        "acm_adam_step: tensor %d has a NULL pointer", first + i);
            ACM_REQUIRE(t.numel < ((int64_t)1 << 40), ACM_EUNSUPPORTED, "acm_adam_step: tensor %d too large", first + i);
            pk.p[i] = t.param, pk.g[i] = t.grad, pk.m[i] = t.exp_avg, pk.v[i] = t.exp_avg_sq, pk.step[i] = t.step;
            pk.numel[i] = (long)t.numel;
            pk.first_block[i] = blocks;
            blocks += adam_blocks((long)t.numel);
        }
        pk.first_block[pk.n] = blocks;
        int64_t* adv = first + PACK >= n_tensors ? cfg->also_advance : nullptr;
        if (blocks > 0 && cfg->arrive) {          // one launch: the last block advances the counters
            hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, pk, hp, cfg->arrive, adv);
        } else {
            if (blocks > 0) hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, pk, hp, (int*)nullptr, (int64_t*)nullptr);
            hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(PACK), 0, s, pk, adv);
        }
        ACM_CHECK_HIP(hipGetLastError());
    }
    if (n_tensors == 0 && cfg->also_advance) {
        AdamPack pk{};
        hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(PACK), 0, s, pk, cfg->also_advance);
        ACM_CHECK_HIP(hipGetLastError());
    }
    return ACM_OK;
}
