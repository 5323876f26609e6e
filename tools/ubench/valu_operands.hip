// Micro-benchmark: does the issue cost of v_fma_f32 / v_fmac_f32 depend on how many of its operands are VGPRs?
// (tv_fir and the oscillator kernel average ~4.2 cycles per VALU instruction, valu_rates measures ~2.4 for
// v_fma_f32 v, v, s, s.)  4 waves per SIMD, 8 independent accumulators per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP 64
template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float x[8], y[8], z[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = threadIdx.x * 1e-3f + a + i;
        y[i] = 1.0f + threadIdx.x * 1e-7f * (i + 1);
        z[i] = 1e-3f * (threadIdx.x & 7) + 1e-4f * i;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) x[i] = __builtin_fmaf(x[i], a, b);                 // 1 VGPR source (+ dst)
                if (OP == 1) x[i] = __builtin_fmaf(y[i], a, x[i]);              // v_fmac: 2 VGPR reads (y, acc), 1 SGPR
                if (OP == 2) x[i] = __builtin_fmaf(y[i], z[i], x[i]);           // v_fmac: 3 VGPR reads
                if (OP == 3) x[i] = __builtin_fmaf(y[i], z[(i + r) & 7], x[i]); // 3 VGPR reads, rotating pairs
                if (OP == 4) x[i] = x[i] * y[i];                                // v_mul 2 VGPR
                if (OP == 5) x[i] = x[i] + y[(i + r) & 7];                      // v_add 2 VGPR
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i] + y[i] + z[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int blocks_per_cu) {
    float* out;
    hipMalloc(&out, 256 * 16 * 256 * 4);
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instrs_per_simd = (double)iters * REP * 8 * blocks_per_cu;
    printf("%-40s waves/SIMD=%d  %.3f ms -> %.2f cyc per wave-instr per SIMD at 2.4 GHz\n", name, blocks_per_cu, ms,
           ms * 1e6 / instrs_per_simd * 2.4);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("fma  v, v, s, s   (1 VGPR read)", w);
        run<1>("fmac v, v, s      (2 VGPR reads)", w);
        run<2>("fmac v, v, v      (3 VGPR reads)", w);
        run<3>("fmac v, v, v      (3 reads, rotating)", w);
        run<4>("mul  v, v, v      (2 VGPR reads)", w);
        run<5>("add  v, v, v      (2 reads, rotating)", w);
    }
    return 0;
}
