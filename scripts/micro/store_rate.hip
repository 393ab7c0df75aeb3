// What does it cost to WRITE the outputs of a row-local kernel (n rows x 64 floats + small per-row side outputs) on gfx950,
// as a function of the lane -> address pattern?  Each variant writes the same 43 MB `out` (n = 168 114 rows) and,
// optionally, the side streams of acm_conv_agg_fwd's row-local stage (head_stats 48 B, att 16 B, two 32-byte copies per row).
//   0  linear: lane l of a wave writes 16 bytes at wave_base + 16 l (1 KB contiguous per instruction)
//   1  transposed-MFMA layout: lane (g, m) writes row m, columns 16 t + 4 g .. + 3, t = 0..3 (16 rows x 64 B per instruction)
//   2  16-lane-group layout: lane (g, m) writes row g (4 rows per step), column m + 16 t as dwords (4 rows x 64 B per instruction)
//   3  = 1 with the side streams
//   4  = 1 with non-temporal stores
//   5  = 1 through LDS: the wave transposes its 16 x 64 tile and writes 1 KB contiguous per instruction
// build + run: hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ out, float* __restrict__ stats, float* __restrict__ att,
                                                    float* __restrict__ c0, float* __restrict__ c1, int n_rows, float seed) {
    __shared__ f32x4 tile[4][16 * 17];
    const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    for (int base = wave * 16; base < n_rows; base += nwaves * 16) {
        const int row = base + m;
        const bool valid = row < n_rows;
        f32x4 v = {seed + lane, seed, seed * 2.f, seed + base};
        if (PAT == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (base + 4 * t + g < n_rows) *reinterpret_cast<f32x4*>(out + (size_t)(base + 4 * t) * 64 + lane * 4) = v;   // rows base+4t .. +3: 1 KB
        } else if (PAT == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = base + 4 * q + g;
                if (r < n_rows) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) out[(size_t)r * 64 + m + 16 * t] = v[t];
                }
            }
        } else if (PAT == 5) {
#pragma unroll
            for (int t = 0; t < 4; ++t) tile[wv][m * 17 + 4 * t + g] = v;          // row m, 16-byte column block 4 t + g
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = 4 * t + g;                                            // row r of the tile, block m
                if (base + r < n_rows) *reinterpret_cast<f32x4*>(out + (size_t)(base + r) * 64 + m * 4) = tile[wv][r * 17 + m];
            }
        } else if (valid) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4* dst = reinterpret_cast<f32x4*>(out + (size_t)row * 64 + 16 * t + 4 * g);
                if (PAT == 4) __builtin_nontemporal_store(v, dst);
                else *dst = v;
            }
            if (PAT == 3) {
                if (g == 0) {
                    reinterpret_cast<f32x4*>(stats + (size_t)row * 12)[0] = v;
                    reinterpret_cast<f32x4*>(stats + (size_t)row * 12)[1] = v;
                    reinterpret_cast<f32x4*>(stats + (size_t)row * 12)[2] = v;
                }
                if (g == 1) *reinterpret_cast<f32x4*>(att + (size_t)row * 4) = v;
                c0[(size_t)row * 8 + g] = v.x;
                c0[(size_t)row * 8 + 4 + g] = v.y;
                c1[(size_t)row * 8 + g] = v.z;
                c1[(size_t)row * 8 + 4 + g] = v.w;
            }
        }
    }
}

__global__ void flush_kernel(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f;
}

int main() {
    const int n = 168114;
    float *out, *stats, *att, *c0, *c1, *junk;
    const size_t junk_n = (size_t)128 << 20;     // 512 MB: beyond the Infinity Cache
    hipMalloc(&out, (size_t)n * 64 * 4 + 4096);
    hipMalloc(&stats, (size_t)n * 12 * 4 + 4096);
    hipMalloc(&att, (size_t)n * 4 * 4 + 4096);
    hipMalloc(&c0, (size_t)n * 8 * 4 + 4096);
    hipMalloc(&c1, (size_t)n * 8 * 4 + 4096);
    hipMalloc(&junk, junk_n * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int grids[] = {512, 1024, 2627};
    for (int pat = 0; pat < 6; ++pat)
        for (int gi = 0; gi < 3; ++gi) {
            float best = 1e9f, sum = 0.f;
            for (int rep = 0; rep < 6; ++rep) {
                hipLaunchKernelGGL(flush_kernel, dim3(4096), dim3(256), 0, 0, junk, junk_n);
                hipEventRecord(a, 0);
                const int grid = grids[gi];
#define L(P) hipLaunchKernelGGL((store_kernel<P>), dim3(grid), dim3(256), 0, 0, out, stats, att, c0, c1, n, 1.0f + rep)
                switch (pat) {
                    case 0: L(0); break;
                    case 1: L(1); break;
                    case 2: L(2); break;
                    case 3: L(3); break;
                    case 4: L(4); break;
                    default: L(5); break;
                }
#undef L
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (rep > 0) {
                    best = ms < best ? ms : best;
                    sum += ms;
                }
            }
            const double mb = (double)n * 64 * 4 / 1e6 + (pat == 3 ? (double)n * (48 + 16 + 64) / 1e6 : 0.0);
            printf("pattern %d grid %4d: best %.1f us, mean %.1f us  (%.0f MB -> %.2f TB/s at best)\n", pat, grids[gi], best * 1e3,
                   sum / 5 * 1e3, mb, mb / best / 1e3);
        }
    return 0;
}
