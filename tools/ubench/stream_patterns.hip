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
    return 0;
}
