// Launch floor of dependent kernel nodes in a captured hipGraph on one stream (MI355X).
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1.f;
}
__global__ void k_chain(const float* p, float* out, int n) {     // one block, serial dependent loads (a final reduce)
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) { if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}

template <class F>
int run(const char* name, int nodes, F launch) {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < nodes; ++i) launch(s, i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const int reps = 20;
    CK(hipEventRecord(a, s));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %7.2f us per node\n", name, ms * 1000.f / (reps * nodes));
    return 0;
}

int main() {
    float *p, *q;
    const int n = 168114 * 64;
    CK(hipMalloc(&p, (size_t)n * 4)); CK(hipMalloc(&q, 4096));
    CK(hipMemset(p, 0, (size_t)n * 4));
    run("empty <<<1,64>>>", 200, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); });
    run("empty <<<657,256>>>", 200, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(657), dim3(256), 0, s); });
    run("empty <<<8192,256>>>", 200, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(8192), dim3(256), 0, s); });
    run("touch 168k floats <<<657,256>>>", 200, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_touch, dim3(657), dim3(256), 0, s, p, 168114); });
    run("touch 336k floats (1.3 MB) <<<1314,256>>>", 200, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_touch, dim3(1314), dim3(256), 0, s, p, 336228); });
    run("touch 10.7M floats (43 MB) <<<42028,256>>>", 100, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_touch, dim3(42028), dim3(256), 0, s, p, n); });
    run("one-block reduce of 1024 partials", 200, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), 0, s, p, q, 1024); });
    run("alternate touch(43 MB) / one-block reduce", 100, [&](hipStream_t s, int i) {
        if (i & 1) hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), 0, s, p, q, 1024);
        else hipLaunchKernelGGL(k_touch, dim3(42028), dim3(256), 0, s, p, n);
    });
    return 0;
}
