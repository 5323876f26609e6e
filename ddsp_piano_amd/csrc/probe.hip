// Measurement aid behind the C-ABI (bench.py's `roofline.measured_peak`): a pure read of a buffer with the access
// pattern of the materialised oscillator bank -- every wavefront streams its own contiguous region, 1 KB per
// instruction, sixteen instructions in flight -- and nothing else.  No arithmetic beyond one add per loaded float, no
// stores: whatever rate this reaches on a buffer is a ceiling for ddspp_cos_oscillator_bank on the same buffer.
// (tools/ubench/stream_patterns.hip is where the pattern was chosen: 6.2-6.4 TB/s of the 8 TB/s on the MI355X.)
#include "ddspp_common.h"

namespace ddspp {

template <int UNROLL>
__global__ void __launch_bounds__(256) hbm_read_streams_kernel(const float* __restrict__ x, size_t per_wave, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const float* p = x + wave * per_wave;
    float acc = 0.f;
    constexpr size_t step = 64 * 4;
    for (size_t i = 0; i < per_wave; i += step * UNROLL) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p + i + u * step) + lane);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    if (acc == 1.2345e30f) sink[0] = acc;        // never true for finite audio-rate data: keeps the loads alive
}

// The write ceiling (round 6; the stand-alone upsamplers are pure write streams): every wavefront fills its own contiguous
// region with 16-byte stores, 1 KB per instruction, non-temporal or plain.
template <bool NT>
__global__ void __launch_bounds__(256) hbm_write_streams_kernel(float* __restrict__ x, size_t per_wave, float v) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v* p = reinterpret_cast<f4v*>(x + wave * per_wave) + lane;
    const f4v val = {v, v + 1.0f, v + 2.0f, v + 3.0f};
    for (size_t i = 0; i < per_wave / 4; i += 64 * 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (NT) __builtin_nontemporal_store(val, p + i + u * 64);
            else p[i + u * 64] = val;
        }
    }
}

// The other ceiling (round 5; bench.py's `roofline_step.measured_ceiling`): what the chip SUSTAINS in wave64 multiply-adds.
// The step's two large kernels are bound by VALU issue, and "one instruction per 2 cycles at 2.4 GHz" is not what a SIMD
// delivers for long: under a pure stream of independent v_fmac_f32 (three VGPR operands, twelve accumulators per lane -- the
// operand mix of the FilteredNoise walk) the socket reaches its power limit and the clock falls (tools/ubench/fma_ceiling:
// 1.04 ns per instruction and SIMD at 2 or 4 wavefronts per SIMD).  iters x 192 multiply-adds per lane.
__global__ void __launch_bounds__(256) fma_stream_kernel(float* __restrict__ sink, int iters, float sa, float sb) {
    float acc[12], tap[16], x[4];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = threadIdx.x * 1e-6f + i;
#pragma unroll
    for (int i = 0; i < 16; ++i) tap[i] = 1e-4f * (float)((threadIdx.x + i) & 31) + sb;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = 1e-3f * (float)((threadIdx.x * 3 + i) & 15) + sa;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int e = 0; e < 12; ++e) acc[e] = __builtin_fmaf(x[d], tap[e - d + 3], acc[e]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i];
    if (s == 1.2345e30f) sink[0] = s;            // keeps the chains alive
}

}  // namespace ddspp

extern "C" {

// Runs `iters` x 192 independent multiply-adds per lane on 256 x waves_per_simd workgroups of four wavefronts (waves_per_simd
// wavefronts on every SIMD of the chip); *wave_fmas_per_simd: wave64 multiply-adds each SIMD executed.  Time it with events:
// ns per instruction and SIMD = elapsed / *wave_fmas_per_simd.  Give it ~20 ms (iters ~ 100 000 at 4 per SIMD): the power
// controller needs milliseconds to settle.
int ddspp_fma_probe(float* sink, int waves_per_simd, int iters, double* wave_fmas_per_simd, hipStream_t stream) {
    DDSPP_REQUIRE(sink && waves_per_simd >= 1 && waves_per_simd <= 8 && iters > 0, "fma_probe: bad arguments");
    hipLaunchKernelGGL(ddspp::fma_stream_kernel, dim3(256 * waves_per_simd), dim3(256), 0, stream, sink, iters, 1.0001f, 0.25f);
    DDSPP_LAUNCH_CHECK();
    if (wave_fmas_per_simd) *wave_fmas_per_simd = (double)iters * 192.0 * waves_per_simd;
    return DDSPP_OK;
}

// Reads the first `n_floats` (rounded down to a whole number of 16 KB wavefront steps) of x with `n_waves` concurrent
// streams (a multiple of 4; 2048 = two per SIMD is what the graded kernel runs); *bytes_read: what was read.
int ddspp_hbm_read_probe(const float* x, size_t n_floats, int n_waves, float* sink, size_t* bytes_read, hipStream_t stream) {
    DDSPP_REQUIRE(x && sink && n_waves >= 4 && n_waves % 4 == 0, "hbm_read_probe: bad arguments");
    DDSPP_REQUIRE((uintptr_t)x % 16 == 0, "hbm_read_probe: the buffer must be 16-byte aligned");
    constexpr size_t chunk = 64 * 4 * 16;                 // floats per wavefront step
    const size_t per_wave = (n_floats / (size_t)n_waves) / chunk * chunk;
    DDSPP_REQUIRE(per_wave > 0, "hbm_read_probe: buffer too small for %d streams", n_waves);
    hipLaunchKernelGGL((ddspp::hbm_read_streams_kernel<16>), dim3(n_waves / 4), dim3(256), 0, stream, x, per_wave, sink);
    DDSPP_LAUNCH_CHECK();
    if (bytes_read) *bytes_read = per_wave * (size_t)n_waves * sizeof(float);
    return DDSPP_OK;
}

// Fills the first `n_floats` (rounded down to whole 8 KB wavefront steps) of x from `n_waves` concurrent streams (a multiple of
// 4), with non-temporal (1) or plain (0) 16-byte stores; *bytes_written: what was written.
int ddspp_hbm_write_probe(float* x, size_t n_floats, int n_waves, int nontemporal, size_t* bytes_written, hipStream_t stream) {
    DDSPP_REQUIRE(x && n_waves >= 4 && n_waves % 4 == 0, "hbm_write_probe: bad arguments");
    DDSPP_REQUIRE((uintptr_t)x % 16 == 0, "hbm_write_probe: the buffer must be 16-byte aligned");
    constexpr size_t chunk = 64 * 4 * 8;                  // floats per wavefront step
    const size_t per_wave = (n_floats / (size_t)n_waves) / chunk * chunk;
    DDSPP_REQUIRE(per_wave > 0, "hbm_write_probe: buffer too small for %d streams", n_waves);
    if (nontemporal) hipLaunchKernelGGL((ddspp::hbm_write_streams_kernel<true>), dim3(n_waves / 4), dim3(256), 0, stream, x, per_wave, 1.0f);
    else hipLaunchKernelGGL((ddspp::hbm_write_streams_kernel<false>), dim3(n_waves / 4), dim3(256), 0, stream, x, per_wave, 1.0f);
    DDSPP_LAUNCH_CHECK();
    if (bytes_written) *bytes_written = per_wave * (size_t)n_waves * sizeof(float);
    return DDSPP_OK;
}

}  // extern "C"
