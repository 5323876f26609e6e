// Micro-benchmark: does a plain VALU instruction issue in the shadow of a quarter-rate transcendental (v_cos_f32) on
// gfx950?  Per repetition: 8 independent v_cos_f32 interleaved with M plain v_fma_f32 each (M = 0..6), all on
// independent registers.  If the transcendental blocks the VALU for its 8 cycles the time is 8 * (8 + 2.3 M) cycles;
// if plain ops slip in underneath it is 8 * max(8, ...) for small M.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP 32
template <int M>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float y[8], x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { y[i] = threadIdx.x * 1e-3f + i; x[i] = y[i] + 0.5f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_cos_f32 %0, %0" : "+v"(y[i]));
#pragma unroll
                for (int m = 0; m < M; ++m) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(i + m) & 7]) : "v"(a), "v"(b));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + y[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int M>
void run(int blocks_per_cu) {
    float* out;
    hipMalloc(&out, 256 * 16 * 256 * 4);
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<M>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<M>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double groups_per_simd = (double)iters * REP * 8 * blocks_per_cu;      // (1 cos + M fma) groups per SIMD
    printf("1 v_cos + %d v_fma   waves/SIMD=%d  %.2f cycles per group at 2.4 GHz (cos alone ~8.1, fma alone ~2.3 each)\n", M,
           blocks_per_cu, ms * 1e6 / groups_per_simd * 2.4);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>(w); run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<6>(w);
    }
    return 0;
}
