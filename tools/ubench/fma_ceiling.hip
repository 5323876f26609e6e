// What does the chip SUSTAIN in float32 multiply-adds?  The VALU-bound kernels of the step (compacted oscillator bank,
// FilteredNoise walk) run with the socket at its 1400 W limit (tools/power_probe.sh), so their ceiling is not "one
// wave64 instruction per 2 cycles at 2.4 GHz" but what the power management lets through.  Streams of independent
// multiply-adds over 12 accumulators, the operand mix of the FilteredNoise walk (acc += x * tap, all three in VGPRs),
// for ~20 ms each (the power controller needs milliseconds to settle), 1 - 4 wavefronts per SIMD:
//   vvv   v_fmac_f32 acc, x, tap          three distinct VGPR reads per instruction
//   svv   v_fmac_f32 acc, s, tap          x from an SGPR
//   pk    v_pk_fma_f32                    the same multiply-adds, two per instruction
//   kvv   v_fmac_f32 acc, x, acc-only     x = a kernel argument (SGPR) and b = literal: one VGPR read (valu_rates' form)
// prints ns per wave64 multiply-add per SIMD, TFLOP/s over the chip, and shader cycles / ns (s_memtime against events).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize fma_ceiling.hip -o fma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) k(float* __restrict__ out, long long* __restrict__ cyc, int iters, float sa, float sb) {
    float acc[12], tap[16], x[4];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = threadIdx.x * 1e-6f + i;
#pragma unroll
    for (int i = 0; i < 16; ++i) tap[i] = 1e-4f * (float)((threadIdx.x + i) & 31) + sb;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = 1e-3f * (float)((threadIdx.x * 3 + i) & 15) + sa;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (OP == 0) {
#pragma unroll
                    for (int e = 0; e < 12; ++e) acc[e] = __builtin_fmaf(x[d], tap[e - d + 3], acc[e]);
                } else if (OP == 1) {
#pragma unroll
                    for (int e = 0; e < 12; ++e) acc[e] = __builtin_fmaf(sa, tap[e - d + 3], acc[e]);
                } else if (OP == 2) {
#pragma unroll
                    for (int e = 0; e < 12; e += 2) {
                        f2 a = {acc[e], acc[e + 1]}, t = {tap[(e + 4 - d) & ~1], tap[((e + 4 - d) & ~1) + 1]}, xx = {x[d], x[d]};
                        a = __builtin_elementwise_fma(xx, t, a);
                        acc[e] = a.x; acc[e + 1] = a.y;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 12; ++e) acc[e] = __builtin_fmaf(acc[e], sa, 0.5f);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, int waves, double target_ms) {
    float* out; long long* cyc;
    const int grid = 256 * waves;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipMalloc(&cyc, (size_t)grid * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {          // first pass sizes the second to ~target_ms
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0001f, 0.25f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (pass == 0) iters = (int)(iters * target_ms / ms) + 1;
    }
    std::vector<long long> h((size_t)grid * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double cs = 0; for (auto v : h) cs += (double)v;
    const double fma_per_simd = (double)iters * 4 * 48 * waves;      // wave64 multiply-adds per SIMD
    const double ns = ms * 1e6 / fma_per_simd;
    printf("%-4s waves/SIMD %d: %7.2f ms  %.3f ns per wave64 multiply-add per SIMD = %6.1f TFLOP/s  (%.2f shader cycles per ns)\n", name,
           waves, ms, ns, 1024.0 * 128.0 / ns * 1e-3, cs / h.size() / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv) {
    const double target = argc > 1 ? atof(argv[1]) : 20.0;
    for (int w : {1, 2, 3, 4}) {
        run<0>("vvv", w, target);
        run<1>("svv", w, target);
        run<2>("pk", w, target);
        run<3>("kvv", w, target);
    }
    return 0;
}
