// Read-bandwidth micro-benchmark for the access patterns of the oscillator-bank kernel:
//   A  flat float4 grid-stride read (the usual roofline probe)
//   B  "row streams": wave w walks its own contiguous region of `row_bytes`, 256 B (dword/lane) or
//      1 KB (dwordx4/lane) per instruction, DEPTH instructions in flight per wave
// 75 GB working set so nothing lives in the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(256) flat_read(const float4* __restrict__ x, size_t n4, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = x[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345f) out[0] = acc;
}

// each wave: stream of `per_wave` floats starting at wave*per_wave; UNROLL dword loads in flight
template <int UNROLL, int VEC>
__global__ void __launch_bounds__(256) row_streams(const float* __restrict__ x, size_t per_wave, float* out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const float* p = x + wave * per_wave;
    float acc = 0.f;
    const size_t step = 64 * VEC;
    for (size_t i = 0; i < per_wave; i += step * UNROLL) {
        float v[UNROLL][VEC];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (VEC == 1) v[u][0] = p[i + u * step + lane];
            else {
                float4 t = *reinterpret_cast<const float4*>(p + i + u * step + lane * 4);
                v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc += v[u][e];
    }
    if (acc == 1.2345f) out[0] = acc;
}

// C  the graded kernel's own pattern: workgroup = one row of [N, 128] floats in each of TWO arrays (fe, ae), G wavefronts
//    per row; wavefront g reads floats 128 / G * g .. of every sample row (G = 2: 256-byte pieces at a 512-byte stride,
//    dword per lane; G = 1: the whole 512-byte row, dwordx2 per lane), SAMPLES x 2 loads in flight per wavefront.
//    SPLIT: wavefront 0 reads only fe, wavefront 1 only ae (whole rows, dwordx2): two sequential streams instead.
typedef float f2v __attribute__((ext_vector_type(2)));
template <int G, int SAMPLES, bool SPLIT>
__global__ void __launch_bounds__(64 * G) bank_pattern(const float* __restrict__ fe, const float* __restrict__ ae, size_t N, float* out) {
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const float* f = fe + (size_t)blockIdx.x * N * 128;
    const float* a = ae + (size_t)blockIdx.x * N * 128;
    float acc = 0.f;
    if (SPLIT) {
        const float* src = g == 0 ? f : a;
        for (size_t n = 0; n < N; n += 2 * SAMPLES) {
            f2v v[2 * SAMPLES];
#pragma unroll
            for (int u = 0; u < 2 * SAMPLES; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(src + (n + u) * 128) + lane);
#pragma unroll
            for (int u = 0; u < 2 * SAMPLES; ++u) acc += v[u].x + v[u].y;
        }
    } else if (G == 2) {
        for (size_t n = 0; n < N; n += SAMPLES) {
            float vf[SAMPLES], va[SAMPLES];
#pragma unroll
            for (int u = 0; u < SAMPLES; ++u) {
                vf[u] = __builtin_nontemporal_load(f + (n + u) * 128 + 64 * g + lane);
                va[u] = __builtin_nontemporal_load(a + (n + u) * 128 + 64 * g + lane);
            }
#pragma unroll
            for (int u = 0; u < SAMPLES; ++u) acc += vf[u] + va[u];
        }
    } else {
        for (size_t n = 0; n < N; n += SAMPLES) {
            f2v vf[SAMPLES], va[SAMPLES];
#pragma unroll
            for (int u = 0; u < SAMPLES; ++u) {
                vf[u] = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(f + (n + u) * 128) + lane);
                va[u] = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(a + (n + u) * 128) + lane);
            }
#pragma unroll
            for (int u = 0; u < SAMPLES; ++u) acc += (vf[u].x + vf[u].y) + (va[u].x + va[u].y);
        }
    }
    if (acc == 1.2345f) out[0] = acc;
}

template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f();
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        f();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t bytes = 72ull << 30;
    float* x; float* out;
    if (hipMalloc(&x, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&out, 4);
    (void)hipMemset(x, 0, bytes);
    const size_t n = bytes / 4;
    float ms = timeit([&] { hipLaunchKernelGGL(flat_read, dim3(256 * 16), dim3(256), 0, 0, (const float4*)x, n / 4, out); });
    printf("flat float4 read                         : %.2f ms  %.0f GB/s\n", ms, bytes / ms / 1e6);
    for (int waves : {1024, 2048, 4096, 8192}) {
        const size_t per_wave = (n / waves) / (64 * 4 * 32) * (64 * 4 * 32);
        const double b = (double)per_wave * waves * 4;
        ms = timeit([&] { hipLaunchKernelGGL((row_streams<16, 1>), dim3(waves / 4), dim3(256), 0, 0, x, per_wave, out); });
        printf("row streams waves=%5d dword  x16 in flight: %.2f ms  %.0f GB/s\n", waves, ms, b / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((row_streams<32, 1>), dim3(waves / 4), dim3(256), 0, 0, x, per_wave, out); });
        printf("row streams waves=%5d dword  x32 in flight: %.2f ms  %.0f GB/s\n", waves, ms, b / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((row_streams<64, 1>), dim3(waves / 4), dim3(256), 0, 0, x, per_wave, out); });
        printf("row streams waves=%5d dword  x64 in flight: %.2f ms  %.0f GB/s\n", waves, ms, b / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((row_streams<8, 4>), dim3(waves / 4), dim3(256), 0, 0, x, per_wave, out); });
        printf("row streams waves=%5d dwordx4 x8 in flight: %.2f ms  %.0f GB/s\n", waves, ms, b / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((row_streams<16, 4>), dim3(waves / 4), dim3(256), 0, 0, x, per_wave, out); });
        printf("row streams waves=%5d dwordx4 x16 in flight: %.2f ms  %.0f GB/s\n", waves, ms, b / ms / 1e6);
    }
    {   // the graded kernel's pattern: 1024 rows x 72000 samples x 128 floats x two arrays = 75.5 GB
        const size_t R = 1024, N = 72000;
        const float* fe = x; const float* ae = x + R * N * 128;
        const double b = (double)R * N * 128 * 4 * 2;
        ms = timeit([&] { hipLaunchKernelGGL((bank_pattern<2, 24, false>), dim3(R), dim3(128), 0, 0, fe, ae, N, out); });
        printf("bank pattern: 2 wavefronts per row, 256-byte halves of both arrays, 48 loads in flight: %.2f ms  %.0f GB/s\n", ms, b / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((bank_pattern<2, 16, false>), dim3(R), dim3(128), 0, 0, fe, ae, N, out); });
        printf("bank pattern: 2 wavefronts per row, 256-byte halves of both arrays, 32 loads in flight: %.2f ms  %.0f GB/s\n", ms, b / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((bank_pattern<1, 24, false>), dim3(R), dim3(64), 0, 0, fe, ae, N, out); });
        printf("bank pattern: 1 wavefront per row, whole 512-byte rows of both arrays, 48 loads in flight: %.2f ms  %.0f GB/s\n", ms, b / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((bank_pattern<2, 24, true>), dim3(R), dim3(128), 0, 0, fe, ae, N, out); });
        printf("bank pattern: 2 wavefronts per row, one array each (whole rows), 48 loads in flight: %.2f ms  %.0f GB/s\n", ms, b / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((bank_pattern<2, 12, true>), dim3(R), dim3(128), 0, 0, fe, ae, N, out); });
        printf("bank pattern: 2 wavefronts per row, one array each (whole rows), 24 loads in flight: %.2f ms  %.0f GB/s\n", ms, b / ms / 1e6);
    }
    return 0;
}
