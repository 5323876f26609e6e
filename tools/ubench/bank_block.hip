// Micro-benchmark: the const-frequency block of bank_compact.hip in isolation (8 samples x 2 oscillators per lane:
// phase scan, + offset, exact 2 pi reduction, v_cos, Hann cross-fade FMA, harmonic sum), 4 wavefronts per SIMD,
// no frame / chunk / tile logic.  Variants: W in SGPRs (as the kernel has them) or VGPRs; staged with sched_barriers
// or left to the compiler.  Reports SIMD cycles per wave64 VALU instruction (144 plain + 16 v_cos per block).
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr float INV_P = 0x1.45f306p-3f;
constexpr float P2 = 6.2831855f;

template <bool STAGED, bool WSGPR, bool NOCOS, bool QBLK = false, bool SPLIT4 = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k(float* out, const float* __restrict__ wtab, int iters, float om0, float om1) {
    typedef const __attribute__((address_space(4))) float* cfloat_p;
    const cfloat_p wc = (cfloat_p)(uintptr_t)wtab;
    float ph[2] = {threadIdx.x * 1e-3f, threadIdx.x * 2e-3f}, off[2] = {0.5f, 1.5f}, om[2] = {om0, om1};
    float da[2] = {0.25f, 0.125f}, a0[2] = {0.5f, 0.25f};
    float total = 0.f;
    for (int it = 0; it < iters; ++it) {
        float w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = WSGPR ? wc[(it & 7) * 8 + i] : wtab[(it & 7) * 8 + i] * (1.0f + threadIdx.x * 1e-9f);
        float pv[8][2], q[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { ph[j] = ph[j] + om[j]; pv[i][j] = ph[j]; }
        if (STAGED) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) pv[i][j] = pv[i][j] + off[j];
        if (QBLK) {
            float q0[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) q0[j] = -__builtin_rintf(pv[0][j] * INV_P);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) pv[i][j] = __builtin_fmaf(q0[j], P2, pv[i][j]);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) pv[i][j] = pv[i][j] * INV_P;
        } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) q[i][j] = __builtin_rintf(pv[i][j] * INV_P);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) pv[i][j] = __builtin_fmaf(-q[i][j], P2, pv[i][j]) * INV_P;
        }
        if (STAGED) __builtin_amdgcn_sched_barrier(0);
        if (!NOCOS) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) pv[i][j] = __builtin_amdgcn_cosf(pv[i][j]);
        }
        if (STAGED) __builtin_amdgcn_sched_barrier(0);
        float acc[8];
        if (SPLIT4) {
            float am[8][2];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) am[i][j] = __builtin_fmaf(da[j], w[i], a0[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = am[i][0] * pv[i][0];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(am[i][1], pv[i][1], acc[i]);
        } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(da[0], w[i], a0[0]) * pv[i][0];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(__builtin_fmaf(da[1], w[i], a0[1]), pv[i][1], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) total += acc[i];
        if (ph[0] > 3000.f) { ph[0] -= 3000.f; ph[1] -= 3000.f; }
    }
    out[blockIdx.x * 64 + threadIdx.x] = total;
}

template <bool STAGED, bool WSGPR, bool NOCOS, bool QBLK = false, bool SPLIT4 = false>
void run(const char* name) {
    float *out, *wtab;
    hipMalloc(&out, 256 * 16 * 64 * 4);
    hipMalloc(&wtab, 64 * 4);
    float h[64];
    for (int i = 0; i < 64; ++i) h[i] = 0.01f * i;
    hipMemcpy(wtab, h, sizeof(h), hipMemcpyHostToDevice);
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<STAGED, WSGPR, NOCOS, QBLK, SPLIT4>), dim3(256 * 16), dim3(64), 0, 0, out, wtab, 10, 0.01f, 0.02f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<STAGED, WSGPR, NOCOS, QBLK, SPLIT4>), dim3(256 * 16), dim3(64), 0, 0, out, wtab, iters, 0.01f, 0.02f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double blocks_per_simd = (double)iters * 4;       // 16 one-wave workgroups per CU = 4 wavefronts per SIMD
    printf("%-52s %7.1f cycles per block per SIMD-wave at 2.4 GHz (ideal 144 x 2.3 + 16 x 8.1 = 461)\n", name,
           ms * 1e6 / blocks_per_simd * 2.4);
    (void)hipFree(out); (void)hipFree(wtab);
}

int main() {
    run<true, true, false>("staged, weights in SGPRs");
    run<true, false, false>("staged, weights in VGPRs");
    run<false, true, false>("compiler order, weights in SGPRs");
    run<false, false, false>("compiler order, weights in VGPRs");
    run<true, true, true>("staged, SGPR weights, no v_cos");
    run<true, false, true>("staged, VGPR weights, no v_cos");
    run<true, true, false, true>("staged, SGPR w, one 2 pi multiple per block (112+16)");
    run<true, true, false, true, true>("  + amplitudes first, then the products");
    run<true, false, false, true, true>("  + the same with weights in VGPRs");
    run<true, true, true, true, true>("  + SGPR w, no v_cos");
    return 0;
}
