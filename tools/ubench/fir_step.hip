// Micro-benchmark: one step of the time-varying FIR inner loop (noise.hip) in isolation -- NR ds_read_b128 (taps + a noise
// block) feeding 64 v_fmac_f32 on 16 accumulators -- at 4 wavefronts per SIMD (four 256-thread workgroups per CU), no
// barriers, no global memory.  How many SIMD cycles does a wavefront's step cost as a function of its LDS reads?
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NR, bool PREFETCH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1e-3f * (float)(i & 63);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    float4 tq[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) tq[q] = make_float4(0.5f, 0.25f, 0.125f, 1.0f);
    // tap reads: one address per 16-lane group (broadcast); noise block: 5 blocks apart per lane (conflict free)
    const float* gp = lds + 16 * (lane >> 4);
    const float* xp = lds + 4096 + 20 * (lane & 15) + 4 * (lane >> 4);
    for (int it = 0; it < iters; ++it) {
        const int o = (it & 31) * 64;
#pragma unroll
        for (int q = 0; q < 6; ++q)
            if (q < NR) tq[q] = *reinterpret_cast<const float4*>((q == 5 ? xp : gp + 4 * q) + o);
        if (PREFETCH) __builtin_amdgcn_sched_barrier(0);
        float tp[20];
#pragma unroll
        for (int q = 0; q < 5; ++q) { tp[4 * q] = tq[q].x; tp[4 * q + 1] = tq[q].y; tp[4 * q + 2] = tq[q].z; tp[4 * q + 3] = tq[q].w; }
        const float xs[4] = {tq[5].x, tq[5].y, tq[5].z, tq[5].w};
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = __builtin_fmaf(xs[d], tp[e - d + 3], acc[e]);
    }
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NR, bool PREFETCH>
void run(const char* name) {
    float* out;
    (void)hipMalloc(&out, 256 * 4 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NR, PREFETCH>), dim3(256 * 4), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NR, PREFETCH>), dim3(256 * 4), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 wavefronts x iters steps
    printf("%-44s %7.1f cycles per wavefront step at 2.4 GHz (64 FMAs x 2.3 = 147)\n", name, ms * 1e-3 * 2.4e9 / (4.0 * iters));
    (void)hipFree(out);
}

int main() {
    run<0, false>("64 FMAs, no LDS read");
    run<1, false>("64 FMAs + 1 ds_read_b128");
    run<2, false>("64 FMAs + 2 ds_read_b128");
    run<4, false>("64 FMAs + 4 ds_read_b128");
    run<6, false>("64 FMAs + 6 ds_read_b128 (the kernel's step)");
    return 0;
}
