// What does a divergent 16-byte load cost the CU's vector-memory path?  One wave instruction = 64 lanes x dwordx4 through
// a buffer descriptor; the lanes' offsets follow a pattern, the table is small enough to sit in L1 (16 KB), in L2 (2 MB) or
// beyond it (64 MB).  Prints CU clocks per wave instruction (all 256 CUs busy, 8 waves per SIMD), i.e. the rate at which
// the texture addresser / L1 retire lane requests.
//
// build + run: hipcc --offload-arch=gfx950 -O3 -o ta_rate ta_rate.hip && ./ta_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Pattern { SAME, CONTIG, PAIRS, QUADS, RANDOM16, ALL_OOB, HALF_OOB_PAIRS, HALF_EXEC_PAIRS, RANDOM8_X2, RANDOM4_X1, PAIRS_X2, N_PATTERNS };
static const char* NAMES[] = {"same address", "contiguous 1 KB", "pairs (32 B rows)", "quads (64 B rows)", "random 16 B",
                              "all out of range", "pairs, half the pairs out of range", "pairs, half the wave masked off",
                              "dwordx2, random 8 B", "dword, random 4 B", "dwordx2 pairs (16 B rows)"};

__device__ __forceinline__ unsigned hash(unsigned a) {
    a *= 0x9E3779B1u;
    return a ^ (a >> 15);
}

template <int PAT>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ table, unsigned bytes, int iters, float* __restrict__ sink) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(table), 0, bytes, 0x00020000);
    const unsigned lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned mask = bytes - 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned seed = gw * 7919u;
    if (PAT == HALF_EXEC_PAIRS && lane >= 32) {
        return;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned k = seed + it * 4 + u;
            unsigned off;
            if (PAT == SAME) off = hash(k) & mask & ~15u;
            else if (PAT == CONTIG) off = ((hash(k) & mask & ~1023u) + lane * 16);
            else if (PAT == PAIRS || PAT == HALF_EXEC_PAIRS) off = (hash(k * 64 + (lane >> 1)) & mask & ~31u) + (lane & 1) * 16;
            else if (PAT == QUADS) off = (hash(k * 64 + (lane >> 2)) & mask & ~63u) + (lane & 3) * 16;
            else if (PAT == RANDOM16) off = hash(k * 64 + lane) & mask & ~15u;
            else if (PAT == ALL_OOB) off = 0xfffffff0u - (lane & 1) * 16;
            else if (PAT == HALF_OOB_PAIRS) off = (lane & 2) ? 0xfffffff0u : (hash(k * 64 + (lane >> 1)) & mask & ~31u) + (lane & 1) * 16;
            else if (PAT == RANDOM8_X2) off = hash(k * 64 + lane) & mask & ~7u;
            else if (PAT == RANDOM4_X1) off = hash(k * 64 + lane) & mask & ~3u;
            else off = (hash(k * 64 + (lane >> 1)) & mask & ~15u) + (lane & 1) * 8;   // PAIRS_X2
            if (PAT == RANDOM8_X2 || PAT == PAIRS_X2) {
                const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
                acc[0] += v[0], acc[1] += v[1];
            } else if (PAT == RANDOM4_X1) {
                acc[0] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
            } else {
                acc += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[gw] = acc[0];
}

template <int PAT>
static void run(const float* table, float* sink, double clk_ghz) {
    static const unsigned sizes[] = {16u << 10, 2u << 20, 64u << 20};
    static const char* where[] = {"L1 (16 KB)", "L2 (2 MB)", "beyond L2 (64 MB)"};
    printf("%-38s", NAMES[PAT]);
    for (int s = 0; s < 3; ++s) {
        const int blocks = 256 * 8, iters = 400;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<PAT>, dim3(blocks), dim3(256), 0, 0, table, sizes[s], iters, sink);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<PAT>, dim3(blocks), dim3(256), 0, 0, table, sizes[s], iters, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_cu = (double)blocks * 4 * iters * 4 / 256.0;
        const double clks = ms * 1e-3 * clk_ghz * 1e9 / instr_per_cu;
        printf("  %-18s %7.1f clk/instr", where[s], clks);
        (void)where;
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double clk_ghz = prop.clockRate * 1e-6;
    printf("%s, %d CUs, %.2f GHz; clocks per 64-lane load instruction and CU\n", prop.name, prop.multiProcessorCount, clk_ghz);
    float *table, *sink;
    hipMalloc(&table, 64u << 20);
    hipMemset(table, 0, 64u << 20);
    hipMalloc(&sink, 1 << 20);
    run<SAME>(table, sink, clk_ghz);
    run<CONTIG>(table, sink, clk_ghz);
    run<QUADS>(table, sink, clk_ghz);
    run<PAIRS>(table, sink, clk_ghz);
    run<RANDOM16>(table, sink, clk_ghz);
    run<ALL_OOB>(table, sink, clk_ghz);
    run<HALF_OOB_PAIRS>(table, sink, clk_ghz);
    run<HALF_EXEC_PAIRS>(table, sink, clk_ghz);
    run<PAIRS_X2>(table, sink, clk_ghz);
    run<RANDOM8_X2>(table, sink, clk_ghz);
    run<RANDOM4_X1>(table, sink, clk_ghz);
    return 0;
}
