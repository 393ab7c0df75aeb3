// What makes a dependent kernel start late inside a replayed hipGraph (MI355X)?  The headline step's timeline shows ~6 us between the
// end of a narrow gather (8192 x 256 threads) and the first wave of agg_bwd16_gather (256 workgroups x 512 threads, 256 VGPRs, 40 KB
// of LDS), and again before the optimizer launch -- and ~0 between the other kernels.  Pairs A -> B in a captured chain: A = a 43 MB
// streaming touch, B = a trivial kernel in different launch shapes; reported: us per pair minus us per A alone.
//   hipcc --offload-arch=gfx950 -O3 -o launch_gap launch_gap.hip && ./launch_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_touch(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1.f;
}
template <int THREADS, int LDS_KB, int WPE>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_b(float* p) {
    __shared__ float lds[LDS_KB * 256 + 1];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += lds[1];
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_b_regs(float* p, int n) {      // keeps ~200 VGPRs alive
    float v[192];
#pragma unroll
    for (int i = 0; i < 192; ++i) v[i] = p[(threadIdx.x + i) % n];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 192; ++i) s += v[i] * v[(i * 7) % 192];
    if (s == 123.f) p[0] = s;
}

template <class F>
float run(int nodes, F launch) {
    hipStream_t s;
    hipStreamCreate(&s);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < nodes; ++i) launch(s, i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    const int reps = 20;
    hipEventRecord(a, s);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}

int main() {
    float* p;
    const int n = 168114 * 64;
    CK(hipMalloc(&p, (size_t)n * 4));
    CK(hipMemset(p, 0, (size_t)n * 4));
    auto A = [&](hipStream_t s) { hipLaunchKernelGGL(k_touch, dim3(42028), dim3(256), 0, s, p, n); };
    const int pairs = 50;
    const float a_alone = run(pairs, [&](hipStream_t s, int) { A(s); }) / pairs;
    printf("A alone (43 MB touch, 42028 x 256)                         %7.2f us\n", a_alone);
    auto pair = [&](const char* name, auto launch_b) {
        const float t = run(2 * pairs, [&](hipStream_t s, int i) { if (i & 1) launch_b(s); else A(s); }) / pairs;
        printf("%-58s %7.2f us on top of A\n", name, t - a_alone);
    };
    pair("B = 256 x 256 threads, no LDS", [&](hipStream_t s) { k_b<256, 1, 4><<<256, 256, 0, s>>>(p); });
    pair("B = 256 x 512 threads, no LDS", [&](hipStream_t s) { k_b<512, 1, 4><<<256, 512, 0, s>>>(p); });
    pair("B = 256 x 512 threads, 40 KB LDS", [&](hipStream_t s) { k_b<512, 40, 4><<<256, 512, 0, s>>>(p); });
    pair("B = 256 x 512 threads, 40 KB LDS, waves_per_eu(2,2)", [&](hipStream_t s) { k_b<512, 40, 2><<<256, 512, 0, s>>>(p); });
    pair("B = 256 x 512 threads, ~200 live VGPRs", [&](hipStream_t s) { k_b_regs<512><<<256, 512, 0, s>>>(p, n); });
    pair("B = 256 x 256 threads, ~200 live VGPRs", [&](hipStream_t s) { k_b_regs<256><<<256, 256, 0, s>>>(p, n); });
    pair("B = 1024 x 256 threads, no LDS", [&](hipStream_t s) { k_b<256, 1, 4><<<1024, 256, 0, s>>>(p); });
    pair("B = 8192 x 256 threads, no LDS", [&](hipStream_t s) { k_b<256, 1, 4><<<8192, 256, 0, s>>>(p); });
    return 0;
}
