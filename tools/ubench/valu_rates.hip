// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the oscillator
// kernel is made of.  One wave per SIMD (256 threads per block, 1 block per CU) and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP 64
template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    float x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    f2 pa = {a, a}, pb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                           x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b); }
            if (OP == 1) { p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb); p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb); }
            if (OP == 2) { x0 = __builtin_amdgcn_cosf(x0); x1 = __builtin_amdgcn_cosf(x1); x2 = __builtin_amdgcn_cosf(x2); x3 = __builtin_amdgcn_cosf(x3);
                           x4 = __builtin_amdgcn_cosf(x4); x5 = __builtin_amdgcn_cosf(x5); x6 = __builtin_amdgcn_cosf(x6); x7 = __builtin_amdgcn_cosf(x7); }
            if (OP == 3) { x0 = __builtin_rintf(x0 * a); x1 = __builtin_rintf(x1 * a); x2 = __builtin_rintf(x2 * a); x3 = __builtin_rintf(x3 * a);
                           x4 = __builtin_rintf(x4 * a); x5 = __builtin_rintf(x5 * a); x6 = __builtin_rintf(x6 * a); x7 = __builtin_rintf(x7 * a); }
            if (OP == 4) { x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a; }
            if (OP == 5) { x0 = (x0 >= b) ? 0.f : x1; x1 = (x1 >= b) ? 0.f : x2; x2 = (x2 >= b) ? 0.f : x3; x3 = (x3 >= b) ? 0.f : x4;
                           x4 = (x4 >= b) ? 0.f : x5; x5 = (x5 >= b) ? 0.f : x6; x6 = (x6 >= b) ? 0.f : x7; x7 = (x7 >= b) ? 0.f : x0; }
            if (OP == 6) { p0 = p0 * pa; p1 = p1 * pa; p2 = p2 * pa; p3 = p3 * pa; }
            if (OP == 7) { p0 = p0 + pa; p1 = p1 + pa; p2 = p2 + pa; p3 = p3 + pa; }
            if (OP == 8) { x0 = x0 + a; x1 = x1 + a; x2 = x2 + a; x3 = x3 + a; x4 = x4 + a; x5 = x5 + a; x6 = x6 + a; x7 = x7 + a; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int OP>
void run(const char* name, int ops_per_rep, int blocks_per_cu) {
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
    double instrs_per_simd = (double)iters * REP * ops_per_rep * blocks_per_cu;   // wave-instructions per SIMD
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (x2.4GHz = %.2f cyc)\n", name, blocks_per_cu, ms,
           ms * 1e6 / instrs_per_simd, ms * 1e6 / instrs_per_simd * 2.4);
    hipFree(out);
}

int main() {
    for (int w : {1, 4}) {
        run<0>("v_fma_f32", 8, w);
        run<1>("v_pk_fma_f32", 4, w);
        run<2>("v_cos_f32", 8, w);
        run<3>("v_mul+v_rndne", 16, w);
        run<4>("v_mul_f32", 8, w);
        run<5>("v_cmp+v_cndmask", 16, w);
        run<6>("v_pk_mul_f32", 4, w);
        run<7>("v_pk_add_f32", 4, w);
        run<8>("v_add_f32", 8, w);
    }
    return 0;
}
