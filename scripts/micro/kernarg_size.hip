// Does the size of a kernel's by-value argument struct cost launch time inside a replayed hipGraph (MI355X)?
// Chains of dependent kernels that read ONE word of their argument struct, struct sizes 16 B .. 3.5 KB, against the same
// kernels taking a POINTER to the struct in device memory.  (Round 6: the rocprofv3 timeline of the headline step shows ~6 us
// gaps in front of exactly the two kernels with the largest argument structs; acm_small.hip's kernels take 1.7 KB by value.)
//   hipcc --offload-arch=gfx950 -O3 -o kernarg_size kernarg_size.hip && ./kernarg_size
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N> struct Args { float* p; int w[N]; };
template <int N> __global__ void k_val(Args<N> a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0] += (float)a.w[N - 1]; }
template <int N> __global__ void k_val_all(Args<N> a) {      // every lane reads a different word: the whole struct is fetched
    int s = 0;
    for (int i = threadIdx.x % N; i < N; i += 64) s += a.w[i];
    if (s == 12345) a.p[1] = 1.f;
}
template <int N> __global__ void k_ptr(const Args<N>* a) { if (threadIdx.x == 0 && blockIdx.x == 0) a->p[0] += (float)a->w[N - 1]; }

template <class F>
int run(const char* name, int nodes, F launch) {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < nodes; ++i) launch(s);
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
    printf("%-64s %7.2f us per node\n", name, ms * 1000.f / (reps * nodes));
    return 0;
}

template <int N> int sweep(float* p, int grid) {
    Args<N> h;
    h.p = p;
    for (int i = 0; i < N; ++i) h.w[i] = i;
    Args<N>* d;
    CK(hipMalloc(&d, sizeof(h)));
    CK(hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice));
    char name[128];
    snprintf(name, sizeof name, "by value, %4zu B, <<<%d,256>>>, one word read", sizeof(h), grid);
    run(name, 200, [&](hipStream_t s) { hipLaunchKernelGGL(k_val<N>, dim3(grid), dim3(256), 0, s, h); });
    snprintf(name, sizeof name, "by value, %4zu B, <<<%d,256>>>, whole struct read", sizeof(h), grid);
    run(name, 200, [&](hipStream_t s) { hipLaunchKernelGGL(k_val_all<N>, dim3(grid), dim3(256), 0, s, h); });
    snprintf(name, sizeof name, "by pointer (device memory), %4zu B, <<<%d,256>>>", sizeof(h), grid);
    run(name, 200, [&](hipStream_t s) { hipLaunchKernelGGL(k_ptr<N>, dim3(grid), dim3(256), 0, s, (const Args<N>*)d); });
    return 0;
}

int main() {
    float* p;
    CK(hipMalloc(&p, 4096));
    CK(hipMemset(p, 0, 4096));
    for (int grid : {1, 340, 2048}) {
        sweep<2>(p, grid);
        sweep<120>(p, grid);
        sweep<440>(p, grid);
        sweep<880>(p, grid);
    }
    return 0;
}
