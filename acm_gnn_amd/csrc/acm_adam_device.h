// The Adam / AdamW element update and its per-launch factors, shared by acm_optim.hip (the multi-tensor update) and
// acm_small.hip (updates applied in the epilogue of the kernel that finishes a gradient).  One definition: both paths must
// stay bit-identical in what they do to an element.
#pragma once
#include <math.h>

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

struct AdamFactors {
    float decay_eff, wd, w1, b2, w2, eps;
    bool decoupled;
    __device__ explicit AdamFactors(const AdamScalars& hp)
        : decay_eff(hp.weight_decay == 0.0 ? 1.0f : (float)(1.0 - hp.lr * hp.weight_decay)), wd((float)hp.weight_decay),
          w1((float)(1.0 - hp.beta1)), b2((float)hp.beta2), w2((float)(1.0 - hp.beta2)), eps((float)hp.eps),
          decoupled(hp.decoupled != 0 || hp.weight_decay == 0.0) {}
};

// the two step-dependent factors of a tensor whose step counter reads `step` BEFORE this update
__device__ __forceinline__ void acm_adam_step_factors(const AdamScalars& hp, float step, float& step_size, float& bc2_sqrt) {
    const double k = (double)step + 1.0;
    step_size = (float)(hp.lr / (1.0 - pow(hp.beta1, k)));
    bc2_sqrt = (float)sqrt(1.0 - pow(hp.beta2, k));
}
