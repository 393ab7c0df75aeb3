// Micro-benchmark: the narrow gather P = A X (32-byte / 16-byte rows) over a per-wave id STREAM instead of the CSR.
//
// Question ("what bounds the narrow gathers"): the CSR kernels take 63 us even with every gathered row in
// L1 -- the chain  item descriptor -> column ids -> rows  is only one step deep per wave, so every step pays the latency of
// the id stream.  Here the ids are laid out in the order a wave consumes them (sliced ELL: 4 rows of similar length per
// wave, 32 neighbours per row and step, padded with an out-of-range sentinel that a buffer load answers with zeros), every
// wave walks ONE contiguous stream, so ids can be requested D steps ahead at 4 registers per step of depth, and the slice
// descriptors (uniform per wave) come through the scalar unit.
//
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o sell_gather.so sell_gather.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ f32x4 ld_row(__amdgpu_buffer_rsrc_t rs, int off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}

// desc: per slice 8 ints {n_steps, out0, out1, out2, out3, -, -, -}; wave_ptr[W] .. wave_ptr[W+1] = the wave's slices;
// wave_step[W] = its first step in the stream.  stream: [step][g][e][u] ints.
// pair form: 32-byte rows, lane (g, e, h) fetches half h of the neighbours e*4 .. e*4+3 of group g.
template <int D, int R>
__global__ __launch_bounds__(256) void sell_gather_pair(const int* __restrict__ stream, const int* __restrict__ wave_ptr,
                                                        const int* __restrict__ wave_step, const int* __restrict__ desc,
                                                        const float* __restrict__ x, unsigned x_bytes,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15, e = gl >> 1, h = gl & 1;
    const int W = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    int s = wave_ptr[W];
    const int s_end = wave_ptr[W + 1];
    if (s >= s_end) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const i32x4* ids = reinterpret_cast<const i32x4*>(stream) + (long)wave_step[W] * 32 + (g * 8 + e);
    i32x4 q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = ids[d * 32];
    ids += D * 32;
    const int hoff = h * 16;
    int total = 0;
    for (int t = s; t < s_end; ++t) total += desc[t * 8];
    int rem = desc[s * 8];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 z[R][4];
    // prologue: rows of the first R-1 steps
#pragma unroll
    for (int r = 0; r < R - 1; ++r) {
        const i32x4 j = q[0];
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) q[d] = q[d + 1];
        q[D - 1] = *ids;
        ids += 32;
        z[r][0] = ld_row(rs, j.x * 32 + hoff);
        z[r][1] = ld_row(rs, j.y * 32 + hoff);
        z[r][2] = ld_row(rs, j.z * 32 + hoff);
        z[r][3] = ld_row(rs, j.w * 32 + hoff);
    }
    for (int t = 0; t < total; t += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // request the rows of step t + r + R - 1 into slot (r + R - 1) % R, consume slot r
            const int wr = (r + R - 1) % R;
            {
                const i32x4 j = q[0];
#pragma unroll
                for (int d = 0; d + 1 < D; ++d) q[d] = q[d + 1];
                q[D - 1] = *ids;
                ids += 32;
                z[wr][0] = ld_row(rs, j.x * 32 + hoff);
                z[wr][1] = ld_row(rs, j.y * 32 + hoff);
                z[wr][2] = ld_row(rs, j.z * 32 + hoff);
                z[wr][3] = ld_row(rs, j.w * 32 + hoff);
            }
            if (R > 1) {
                acc += (z[r][0] + z[r][1]) + (z[r][2] + z[r][3]);
            }
            if (R == 1) {
                acc += (z[0][0] + z[0][1]) + (z[0][2] + z[0][3]);
            }
            if (--rem == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] += dpp<0x4E>(acc[i]);
                    acc[i] += dpp<0x124>(acc[i]);
                    acc[i] += dpp<0x128>(acc[i]);
                }
                const int o = desc[s * 8 + 1 + g];
                if (gl < 2 && o >= 0) *reinterpret_cast<f32x4*>(out + (long)o * 8 + 4 * h) = acc;
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
                ++s;
                rem = s < s_end ? desc[s * 8] : 0x7fffffff;
            }
        }
    }
}

// quad form: 16-byte rows, lane (g, gl) fetches neighbours gl*2, gl*2+1 of group g.
template <int D, int R>
__global__ __launch_bounds__(256) void sell_gather_quad(const int* __restrict__ stream, const int* __restrict__ wave_ptr,
                                                        const int* __restrict__ wave_step, const int* __restrict__ desc,
                                                        const float* __restrict__ x, unsigned x_bytes,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15;
    const int W = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    int s = wave_ptr[W];
    const int s_end = wave_ptr[W + 1];
    if (s >= s_end) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const int2* ids = reinterpret_cast<const int2*>(stream) + (long)wave_step[W] * 64 + (g * 16 + gl);
    int2 q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = ids[d * 64];
    ids += D * 64;
    int total = 0;
    for (int t = s; t < s_end; ++t) total += desc[t * 8];
    int rem = desc[s * 8];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 z[R][2];
#pragma unroll
    for (int r = 0; r < R - 1; ++r) {
        const int2 j = q[0];
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) q[d] = q[d + 1];
        q[D - 1] = *ids;
        ids += 64;
        z[r][0] = ld_row(rs, j.x * 16);
        z[r][1] = ld_row(rs, j.y * 16);
    }
    for (int t = 0; t < total; t += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int wr = (r + R - 1) % R;
            {
                const int2 j = q[0];
#pragma unroll
                for (int d = 0; d + 1 < D; ++d) q[d] = q[d + 1];
                q[D - 1] = *ids;
                ids += 64;
                z[wr][0] = ld_row(rs, j.x * 16);
                z[wr][1] = ld_row(rs, j.y * 16);
            }
            acc += z[r][0] + z[r][1];
            if (--rem == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] += dpp<0xB1>(acc[i]);     // quad_perm [1,0,3,2]
                    acc[i] += dpp<0x4E>(acc[i]);
                    acc[i] += dpp<0x124>(acc[i]);
                    acc[i] += dpp<0x128>(acc[i]);
                }
                const int o = desc[s * 8 + 1 + g];
                if (gl == 0 && o >= 0) *reinterpret_cast<f32x4*>(out + (long)o * 4) = acc;
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
                ++s;
                rem = s < s_end ? desc[s * 8] : 0x7fffffff;
            }
        }
    }
}

#define LAUNCH(KERN, DD, RR)                                                                                     \
    if (depth == DD && rows == RR) {                                                                             \
        hipLaunchKernelGGL((KERN<DD, RR>), dim3(n_waves / 4), dim3(256), 0, (hipStream_t)stream_handle, stream,  \
                           wave_ptr, wave_step, desc, x, x_bytes, out);                                          \
        return (int)hipGetLastError();                                                                           \
    }

extern "C" int sell_gather(int width, int depth, int rows, const int* stream, const int* wave_ptr, const int* wave_step,
                           const int* desc, const float* x, unsigned x_bytes, float* out, int n_waves,
                           void* stream_handle) {
    if (width == 8) {
        LAUNCH(sell_gather_pair, 1, 1) LAUNCH(sell_gather_pair, 2, 1) LAUNCH(sell_gather_pair, 4, 1)
        LAUNCH(sell_gather_pair, 2, 2) LAUNCH(sell_gather_pair, 4, 2) LAUNCH(sell_gather_pair, 6, 2)
        LAUNCH(sell_gather_pair, 4, 3) LAUNCH(sell_gather_pair, 8, 2)
    } else if (width == 4) {
        LAUNCH(sell_gather_quad, 1, 1) LAUNCH(sell_gather_quad, 2, 1) LAUNCH(sell_gather_quad, 4, 1)
        LAUNCH(sell_gather_quad, 2, 2) LAUNCH(sell_gather_quad, 4, 2) LAUNCH(sell_gather_quad, 6, 2)
        LAUNCH(sell_gather_quad, 4, 3) LAUNCH(sell_gather_quad, 8, 2)
    }
    return -1;
}

// ---- variants for the probe --------------------------------------------------------------------------------------
// (a) cache-policy bits on the row loads (AUX: 1 = sc0, 2 = nt, 16 = sc1): does skipping the L1 change the per-row cost?
// (b) the first HUB rows of X staged in LDS (one 1024-thread workgroup per CU, up to 128 KB): neighbours with id < HUB
//     are read with ds_read_b128, the rest through the buffer load (complementary exec masks inside one step).
template <int AUX>
__device__ __forceinline__ f32x4 ld_row_aux(__amdgpu_buffer_rsrc_t rs, int off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX));
}

template <int D, int AUX, bool LDSHUB, int WPB>
__global__ __launch_bounds__(WPB * 64) void sell_gather_pair_v(const int* __restrict__ stream, const int* __restrict__ wave_ptr,
                                                               const int* __restrict__ wave_step, const int* __restrict__ desc,
                                                               const float* __restrict__ x, unsigned x_bytes,
                                                               float* __restrict__ out, int hub_rows) {
    extern __shared__ f32x4 hub[];
    if (LDSHUB) {
        for (int i = threadIdx.x; i < hub_rows * 2; i += WPB * 64) hub[i] = reinterpret_cast<const f32x4*>(x)[i];
        if (threadIdx.x < 2) hub[hub_rows * 2 + threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15, e = gl >> 1, h = gl & 1;
    const int W = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + (threadIdx.x >> 6));
    int s = wave_ptr[W];
    const int s_end = wave_ptr[W + 1];
    if (s >= s_end) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const i32x4* ids = reinterpret_cast<const i32x4*>(stream) + (long)wave_step[W] * 32 + (g * 8 + e);
    i32x4 q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = ids[d * 32];
    ids += D * 32;
    const int hoff = h * 16;
    int total = 0;
    for (int t = s; t < s_end; ++t) total += desc[t * 8];
    int rem = desc[s * 8];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < total; ++t) {
        const i32x4 j = q[0];
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) q[d] = q[d + 1];
        q[D - 1] = *ids;
        ids += 32;
        f32x4 z[4];
        if (LDSHUB) {
            // branch-free: hub neighbours read LDS and send an out-of-range offset to the buffer load (answered with
            // zeros), the others read the zero row behind the hub table (one address: an LDS broadcast)
            f32x4 zl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in_hub = j[u] < hub_rows;
                zl[u] = hub[(in_hub ? j[u] : hub_rows) * 2 + h];
                z[u] = ld_row_aux<AUX>(rs, in_hub ? -16 : j[u] * 32 + hoff);
            }
            acc += ((zl[0] + zl[1]) + (zl[2] + zl[3])) + ((z[0] + z[1]) + (z[2] + z[3]));
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) z[u] = ld_row_aux<AUX>(rs, j[u] * 32 + hoff);
            acc += (z[0] + z[1]) + (z[2] + z[3]);
        }
        if (--rem == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] += dpp<0x4E>(acc[i]);
                acc[i] += dpp<0x124>(acc[i]);
                acc[i] += dpp<0x128>(acc[i]);
            }
            const int o = desc[s * 8 + 1 + g];
            if (gl < 2 && o >= 0) *reinterpret_cast<f32x4*>(out + (long)o * 8 + 4 * h) = acc;
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            ++s;
            rem = s < s_end ? desc[s * 8] : 0x7fffffff;
        }
    }
}

#define LAUNCH_V(AUXV, HUBV, WPBV)                                                                                 \
    if (aux == AUXV && (hub_rows > 0) == HUBV && wpb == WPBV) {                                                    \
        auto k = sell_gather_pair_v<2, AUXV, HUBV, WPBV>;                                                          \
        if (HUBV) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, hub_rows * 32 + 32);  \
        hipLaunchKernelGGL(k, dim3(n_waves / WPBV), dim3(WPBV * 64), HUBV ? hub_rows * 32 + 32 : 0,                     \
                           (hipStream_t)stream_handle, stream, wave_ptr, wave_step, desc, x, x_bytes, out, hub_rows); \
        return (int)hipGetLastError();                                                                             \
    }

extern "C" int sell_gather_v(int aux, int hub_rows, int wpb, const int* stream, const int* wave_ptr, const int* wave_step,
                             const int* desc, const float* x, unsigned x_bytes, float* out, int n_waves,
                             void* stream_handle) {
    LAUNCH_V(0, false, 4) LAUNCH_V(1, false, 4) LAUNCH_V(2, false, 4) LAUNCH_V(3, false, 4) LAUNCH_V(16, false, 4)
    LAUNCH_V(17, false, 4) LAUNCH_V(0, true, 16) LAUNCH_V(0, true, 8) LAUNCH_V(0, false, 16)
    return -1;
}

// (c) cache policy of the ID stream: the ids are read once; if they allocate in the L2 like everything else they push
//     the gathered table (5.4 MB against 4 MB of L2 per XCD) out of it.  IDP: 0 plain, 1 nt, 2 sc1, 3 nt + sc1, 4 sc0 + sc1.
template <int IDP>
__device__ __forceinline__ i32x4 ld_ids(__amdgpu_buffer_rsrc_t rs, int off) {
    constexpr int aux = IDP == 0 ? 0 : (IDP == 1 ? 2 : (IDP == 2 ? 16 : (IDP == 3 ? 18 : 17)));
    return __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, aux));
}

template <int D, int IDP>
__global__ __launch_bounds__(256) void sell_gather_pair_id(const int* __restrict__ stream, unsigned stream_bytes,
                                                           const int* __restrict__ wave_ptr,
                                                           const int* __restrict__ wave_step, const int* __restrict__ desc,
                                                           const float* __restrict__ x, unsigned x_bytes,
                                                           float* __restrict__ out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15, e = gl >> 1, h = gl & 1;
    const int W = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    int s = wave_ptr[W];
    const int s_end = wave_ptr[W + 1];
    if (s >= s_end) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(stream), 0, stream_bytes, 0x00020000);
    int ioff = wave_step[W] * 512 + (g * 8 + e) * 16;
    i32x4 q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = ld_ids<IDP>(ri, ioff + d * 512);
    ioff += D * 512;
    const int hoff = h * 16;
    int total = 0;
    for (int t = s; t < s_end; ++t) total += desc[t * 8];
    int rem = desc[s * 8];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < total; ++t) {
        const i32x4 j = q[0];
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) q[d] = q[d + 1];
        q[D - 1] = ld_ids<IDP>(ri, ioff);
        ioff += 512;
        f32x4 z[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) z[u] = ld_row(rs, j[u] * 32 + hoff);
        acc += (z[0] + z[1]) + (z[2] + z[3]);
        if (--rem == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] += dpp<0x4E>(acc[i]);
                acc[i] += dpp<0x124>(acc[i]);
                acc[i] += dpp<0x128>(acc[i]);
            }
            const int o = desc[s * 8 + 1 + g];
            if (gl < 2 && o >= 0) *reinterpret_cast<f32x4*>(out + (long)o * 8 + 4 * h) = acc;
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            ++s;
            rem = s < s_end ? desc[s * 8] : 0x7fffffff;
        }
    }
}

#define LAUNCH_ID(DD, PP)                                                                                             \
    if (depth == DD && idp == PP) {                                                                                   \
        hipLaunchKernelGGL((sell_gather_pair_id<DD, PP>), dim3(n_waves / 4), dim3(256), 0, (hipStream_t)stream_handle, \
                           stream, stream_bytes, wave_ptr, wave_step, desc, x, x_bytes, out);                         \
        return (int)hipGetLastError();                                                                                \
    }

extern "C" int sell_gather_id(int depth, int idp, const int* stream, unsigned stream_bytes, const int* wave_ptr,
                              const int* wave_step, const int* desc, const float* x, unsigned x_bytes, float* out,
                              int n_waves, void* stream_handle) {
    LAUNCH_ID(2, 0) LAUNCH_ID(2, 1) LAUNCH_ID(2, 2) LAUNCH_ID(2, 3) LAUNCH_ID(2, 4)
    LAUNCH_ID(6, 0) LAUNCH_ID(6, 1) LAUNCH_ID(6, 2) LAUNCH_ID(6, 3) LAUNCH_ID(6, 4)
    LAUNCH_ID(12, 1) LAUNCH_ID(12, 2)
    return -1;
}

// (d) the COLD neighbours (column id >= hot_rows) partitioned by column range over the eight XCDs: block b serves range
//     b % 8 (its XCD: the slice of the table it needs, 1/8 of the cold part, stays in that XCD's L2), one lane PAIR per work
//     item (a row's few cold neighbours inside the range), 32 items per wave and slice, ids stored [step][32 items].
//     wave_ptr / wave_step / desc as above (desc: 40 ints per slice = {steps, -, -, -, -, -, -, -, out index x 32}).
__global__ __launch_bounds__(256) void cold_partials(const int* __restrict__ stream, const int* __restrict__ wave_ptr,
                                                     const int* __restrict__ wave_step, const int* __restrict__ desc,
                                                     const float* __restrict__ x, unsigned x_bytes, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, pr = lane >> 1, h = lane & 1;
    const int W = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    int s = wave_ptr[W];
    const int s_end = wave_ptr[W + 1];
    if (s >= s_end) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const int* ids = stream + (long)wave_step[W] * 32 + pr;
    int q0 = ids[0], q1 = ids[32];
    ids += 64;
    const int hoff = h * 16;
    for (; s < s_end; ++s) {
        const int ns = desc[s * 40];
        const int o = desc[s * 40 + 8 + pr];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < ns; ++t) {
            const f32x4 z = ld_row(rs, q0 * 32 + hoff);
            q0 = q1;
            q1 = *ids;
            ids += 32;
            acc += z;
        }
        if (o >= 0) *reinterpret_cast<f32x4*>(out + (long)o * 8 + 4 * h) = acc;
    }
}

extern "C" int cold_gather(const int* stream, const int* wave_ptr, const int* wave_step, const int* desc, const float* x,
                           unsigned x_bytes, float* out, int n_waves, void* stream_handle) {
    hipLaunchKernelGGL(cold_partials, dim3(n_waves / 4), dim3(256), 0, (hipStream_t)stream_handle, stream, wave_ptr, wave_step,
                       desc, x, x_bytes, out);
    return (int)hipGetLastError();
}

// (e) 16-byte rows (the output-layer gathers): the first hub_rows rows of X in LDS (up to 128 KB = 8 192 rows, half of the
//     twitch-shaped graph's edges), branch-free as in (b): hub neighbours read LDS and send an out-of-range offset to the
//     buffer load, the others read the zero row behind the hub table.
template <int D, bool LDSHUB, int WPB>
__global__ __launch_bounds__(WPB * 64) void sell_gather_quad_v(const int* __restrict__ stream, const int* __restrict__ wave_ptr,
                                                               const int* __restrict__ wave_step, const int* __restrict__ desc,
                                                               const float* __restrict__ x, unsigned x_bytes,
                                                               float* __restrict__ out, int hub_rows) {
    extern __shared__ f32x4 hub[];
    if (LDSHUB) {
        for (int i = threadIdx.x; i < hub_rows; i += WPB * 64) hub[i] = reinterpret_cast<const f32x4*>(x)[i];
        if (threadIdx.x == 0) hub[hub_rows] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15;
    const int W = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + (threadIdx.x >> 6));
    int s = wave_ptr[W];
    const int s_end = wave_ptr[W + 1];
    if (s >= s_end) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const int2* ids = reinterpret_cast<const int2*>(stream) + (long)wave_step[W] * 64 + (g * 16 + gl);
    int2 q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = ids[d * 64];
    ids += D * 64;
    int total = 0;
    for (int t = s; t < s_end; ++t) total += desc[t * 8];
    int rem = desc[s * 8];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < total; ++t) {
        const int2 j = q[0];
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) q[d] = q[d + 1];
        q[D - 1] = *ids;
        ids += 64;
        if (LDSHUB) {
            const bool h0 = j.x < hub_rows, h1 = j.y < hub_rows;
            const f32x4 l0 = hub[h0 ? j.x : hub_rows], l1 = hub[h1 ? j.y : hub_rows];
            const f32x4 z0 = ld_row(rs, h0 ? -16 : j.x * 16), z1 = ld_row(rs, h1 ? -16 : j.y * 16);
            acc += (l0 + l1) + (z0 + z1);
        } else {
            acc += ld_row(rs, j.x * 16) + ld_row(rs, j.y * 16);
        }
        if (--rem == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] += dpp<0xB1>(acc[i]);
                acc[i] += dpp<0x4E>(acc[i]);
                acc[i] += dpp<0x124>(acc[i]);
                acc[i] += dpp<0x128>(acc[i]);
            }
            const int o = desc[s * 8 + 1 + g];
            if (gl == 0 && o >= 0) *reinterpret_cast<f32x4*>(out + (long)o * 4) = acc;
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            ++s;
            rem = s < s_end ? desc[s * 8] : 0x7fffffff;
        }
    }
}

#define LAUNCH_Q(HUBV, WPBV)                                                                                       \
    if ((hub_rows > 0) == HUBV && wpb == WPBV) {                                                                   \
        auto k = sell_gather_quad_v<2, HUBV, WPBV>;                                                                \
        if (HUBV) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, hub_rows * 16 + 16); \
        hipLaunchKernelGGL(k, dim3(n_waves / WPBV), dim3(WPBV * 64), HUBV ? hub_rows * 16 + 16 : 0,                \
                           (hipStream_t)stream_handle, stream, wave_ptr, wave_step, desc, x, x_bytes, out, hub_rows); \
        return (int)hipGetLastError();                                                                             \
    }

extern "C" int sell_gather_q(int hub_rows, int wpb, const int* stream, const int* wave_ptr, const int* wave_step,
                             const int* desc, const float* x, unsigned x_bytes, float* out, int n_waves,
                             void* stream_handle) {
    LAUNCH_Q(false, 4) LAUNCH_Q(false, 16) LAUNCH_Q(true, 16) LAUNCH_Q(true, 8)
    return -1;
}

// (f) TWO ROLES IN ONE KERNEL: can a VALU-bound row-local kernel (the layer-1 backward agg_bwd, 151 VGPRs, 3 waves/SIMD,
//     65 us) carry the memory-bound gather of the NEXT step's P = A dropout(X) (it does not depend on the parameter update)
//     in extra waves of the same workgroups?  Two kernels on two streams do not share the CUs (scripts/probe_overlap.py);
//     here waves 0..3 of a block run a dependent-FMA loop (`iters` x 16 FMAs: the stand-in for the backward's instruction
//     stream), waves 4..4+GW-1 walk an id stream each (the pair form above, R steps of rows in flight).  80 KB of dynamic
//     LDS per block keep two blocks on a CU, i.e. (4 + GW) * 2 waves per CU -- the occupancy the backward's registers allow.
//     mode 1: VALU waves only, 2: gather waves only, 3: both.
template <int R, int GW>
__global__ __launch_bounds__((4 + GW) * 64) void roles_kernel(int mode, int iters, const int* __restrict__ stream,
                                                              const int* __restrict__ wave_ptr, const int* __restrict__ wave_step,
                                                              const int* __restrict__ desc, const float* __restrict__ x,
                                                              unsigned x_bytes, float* __restrict__ out, float* __restrict__ sink) {
    extern __shared__ float pad_lds[];
    const int wv = threadIdx.x >> 6;
    if (wv < 4) {
        if (!(mode & 1)) return;
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = (float)(threadIdx.x + k) * 1e-3f;
        const float c1 = 0.999f + 1e-9f * (float)blockIdx.x, c2 = 1e-4f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = __builtin_fmaf(a[k], c1, c2);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += a[k];
        if (s == 123.456f) sink[threadIdx.x] = s + pad_lds[threadIdx.x];
        return;
    }
    if (!(mode & 2)) return;
    constexpr int D = 2;
    const int lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15, e = gl >> 1, h = gl & 1;
    const int W = __builtin_amdgcn_readfirstlane(blockIdx.x * GW + (wv - 4));
    int s = wave_ptr[W];
    const int s_end = wave_ptr[W + 1];
    if (s >= s_end) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const i32x4* ids = reinterpret_cast<const i32x4*>(stream) + (long)wave_step[W] * 32 + (g * 8 + e);
    i32x4 q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = ids[d * 32];
    ids += D * 32;
    const int hoff = h * 16;
    int total = 0;
    for (int t = s; t < s_end; ++t) total += desc[t * 8];
    int rem = desc[s * 8];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 z[R][4];
#pragma unroll
    for (int r = 0; r < R - 1; ++r) {
        const i32x4 j = q[0];
        q[0] = q[1];
        q[1] = *ids;
        ids += 32;
        z[r][0] = ld_row(rs, j.x * 32 + hoff);
        z[r][1] = ld_row(rs, j.y * 32 + hoff);
        z[r][2] = ld_row(rs, j.z * 32 + hoff);
        z[r][3] = ld_row(rs, j.w * 32 + hoff);
    }
    for (int t = 0; t < total; t += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int wr = (r + R - 1) % R;
            {
                const i32x4 j = q[0];
                q[0] = q[1];
                q[1] = *ids;
                ids += 32;
                z[wr][0] = ld_row(rs, j.x * 32 + hoff);
                z[wr][1] = ld_row(rs, j.y * 32 + hoff);
                z[wr][2] = ld_row(rs, j.z * 32 + hoff);
                z[wr][3] = ld_row(rs, j.w * 32 + hoff);
            }
            acc += (z[r][0] + z[r][1]) + (z[r][2] + z[r][3]);
            if (--rem == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] += dpp<0x4E>(acc[i]);
                    acc[i] += dpp<0x124>(acc[i]);
                    acc[i] += dpp<0x128>(acc[i]);
                }
                const int o = desc[s * 8 + 1 + g];
                if (gl < 2 && o >= 0) *reinterpret_cast<f32x4*>(out + (long)o * 8 + 4 * h) = acc;
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
                ++s;
                rem = s < s_end ? desc[s * 8] : 0x7fffffff;
            }
        }
    }
}

#define LAUNCH_R(RR, GG)                                                                                           \
    if (rows == RR && gw == GG) {                                                                                  \
        auto k = roles_kernel<RR, GG>;                                                                             \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);          \
        hipLaunchKernelGGL(k, dim3(n_blocks), dim3((4 + GG) * 64), lds_bytes, (hipStream_t)stream_handle, mode, iters, \
                           stream, wave_ptr, wave_step, desc, x, x_bytes, out, sink);                              \
        return (int)hipGetLastError();                                                                             \
    }

extern "C" int roles(int mode, int rows, int gw, int iters, int n_blocks, int lds_bytes, const int* stream,
                     const int* wave_ptr, const int* wave_step, const int* desc, const float* x, unsigned x_bytes,
                     float* out, float* sink, void* stream_handle) {
    LAUNCH_R(2, 2) LAUNCH_R(4, 2) LAUNCH_R(6, 2) LAUNCH_R(8, 2)
    LAUNCH_R(2, 4) LAUNCH_R(4, 4) LAUNCH_R(6, 4)
    LAUNCH_R(2, 8) LAUNCH_R(4, 8)
    return -1;
}
