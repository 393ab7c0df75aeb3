// Fused Adam / AdamW update over a list of parameter tensors (gfx950).
// torch.optim's multi-tensor path issues ~3 launches per parameter for the bias-correction scalars on top of
// the foreach kernels (~80 launches, ~0.37 ms per step for the 26 parameters of the two-layer model, as much
// as the two sparse products of a layer together).  Here the whole update is one launch per 32 tensors: the
// tensor table travels by value in the kernel arguments (so it is capture-safe: nothing is read from host
// memory at replay), each 256-thread block owns 2048 consecutive elements of one tensor, and the step
// counters live on the device (one fp32 scalar per tensor, the layout torch.optim uses when capturable=True).
#include <math.h>

#include "acm_common.h"

namespace {

constexpr int PACK = 32;          // tensors per launch
constexpr int CHUNK = 2048;       // elements per block

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

struct AdamScalars {
    double lr, beta1, beta2, eps, weight_decay;
    int decoupled;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float decay, float wd, bool decoupled,
                                         float w1, float b2, float w2, float step_size, float bc2_sqrt, float eps) {
    if (decoupled) p *= decay;
    else g = g + wd * p;
    m = m + w1 * (g - m);
    v = v * b2 + w2 * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamPack pk, AdamScalars hp, int* arrive, int64_t* also_advance) {
    __shared__ float sc[2];
    int t = 0;
    while (t + 1 < pk.n && (int)blockIdx.x >= pk.first_block[t + 1]) ++t;      // uniform: <= 31 scalar compares
    if (threadIdx.x == 0) {
        const double k = (double)pk.step[t][0] + 1.0;
        sc[0] = (float)(hp.lr / (1.0 - pow(hp.beta1, k)));
        sc[1] = (float)sqrt(1.0 - pow(hp.beta2, k));
    }
    __syncthreads();
    const float step_size = sc[0], bc2_sqrt = sc[1];
    const float decay = (float)(1.0 - hp.lr * hp.weight_decay), wd = (float)hp.weight_decay;
    const bool decoupled = hp.decoupled != 0 || hp.weight_decay == 0.0;
    const float decay_eff = hp.weight_decay == 0.0 ? 1.0f : decay;
    const float w1 = (float)(1.0 - hp.beta1), b2 = (float)hp.beta2, w2 = (float)(1.0 - hp.beta2), eps = (float)hp.eps;
    float* __restrict__ p = pk.p[t];
    const float* __restrict__ g = pk.g[t];
    float* __restrict__ m = pk.m[t];
    float* __restrict__ v = pk.v[t];
    const long n = pk.numel[t];
    const long base = (long)((int)blockIdx.x - pk.first_block[t]) * CHUNK;
    const long end = base + CHUNK < n ? base + CHUNK : n;
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && end - base == CHUNK;
    if (vec) {
#pragma unroll
        for (int r = 0; r < CHUNK / 1024; ++r) {
            const long i = base + r * 1024 + threadIdx.x * 4;
            float4 pp = *reinterpret_cast<const float4*>(p + i), gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<const float4*>(m + i), vv = *reinterpret_cast<const float4*>(v + i);
            adam_one(pp.x, gg.x, mm.x, vv.x, decay_eff, wd, decoupled, w1, b2, w2, step_size, bc2_sqrt, eps);
            adam_one(pp.y, gg.y, mm.y, vv.y, decay_eff, wd, decoupled, w1, b2, w2, step_size, bc2_sqrt, eps);
            adam_one(pp.z, gg.z, mm.z, vv.z, decay_eff, wd, decoupled, w1, b2, w2, step_size, bc2_sqrt, eps);
            adam_one(pp.w, gg.w, mm.w, vv.w, decay_eff, wd, decoupled, w1, b2, w2, step_size, bc2_sqrt, eps);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        }
    } else {
        for (long i = base + threadIdx.x; i < end; i += 256) {
            float pp = p[i], mm = m[i], vv = v[i];
            adam_one(pp, g[i], mm, vv, decay_eff, wd, decoupled, w1, b2, w2, step_size, bc2_sqrt, eps);
            p[i] = pp;
            m[i] = mm;
            v[i] = vv;
        }
    }
    // Advance the step counters in the same launch: every block read its counter at the top, so the last block to get
    // here may increment them.  Only the arrival counter is shared between blocks (a device-scope atomic; no data is
    // handed from block to block, hence no fence); the incremented values are for the NEXT launch, which the kernel
    // boundary orders.
    if (arrive) {
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
}

// after the update of a pack: step_t += 1 for each of its tensors (stream order makes every block of the update
// read the old value)
__global__ void adam_advance_kernel(AdamPack pk, int64_t* also_advance) {
    const int t = threadIdx.x;
    if (t < pk.n) pk.step[t][0] += 1.0f;
    if (t == 0 && also_advance) also_advance[0] += 1;
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
            blocks += (int)((t.numel + CHUNK - 1) / CHUNK);
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
