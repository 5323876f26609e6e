// Would the FilteredNoise walk be cheaper on the matrix pipe?  A convolution is a sum of outer products: with
// A = 4 taps, B = 4 input samples, v_mfma_f32_4x4x1_16b_f32 adds 16 independent 4 x 4 outer products (256
// multiply-adds, none of them wasted on the ragged ends a vector walk has) per instruction.  On paper the f32 matrix
// rate equals the f32 vector rate (one 256-MAC instruction per 8 cycles against one 64-MAC v_fmac per 2); what
// this measures is what the chip SUSTAINS (the vector walk runs the socket into its 1400 W limit,
// tools/ubench/fma_ceiling) and what operand delivery costs:
//   mfma     independent v_mfma_f32_4x4x1 over NACC accumulators, operands resident in VGPRs
//   mfma+lds the same with the operands of every 4 instructions fetched by two ds_read_b128 (what a walk would do)
//   mix      1 mfma : 4 v_fmac in one instruction stream (do the two pipes add up?)
//   vfma     the vector stream of fma_ceiling (vvv), for the same box
// prints ns per instruction per SIMD and the multiply-add rate over the chip.  argv[1] = ms per measurement (20),
// argv[2] = 1: loop the mfma+lds kernel forever-ish (for tools/power_probe.sh).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize mfma_ceiling.hip -o mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int OP, int NACC>
__global__ void __launch_bounds__(256) k(float* __restrict__ out, int iters, float sa, float sb) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1e-4f * (float)(i & 63) + sb;
    __syncthreads();
    f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f4{threadIdx.x * 1e-6f + i, 0.f, 1.f, 2.f};
    float a[4], b[4], v[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = 1e-4f * (float)((threadIdx.x + i) & 31) + sb;
        b[i] = 1e-3f * (float)((threadIdx.x * 3 + i) & 15) + sa;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 1e-6f + i;
    const float* pa = lds + (threadIdx.x & 63) * 4;
    const float* pb = lds + 1024 + (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP == 1) {          // operands of the next four instructions
                const f4 na = *reinterpret_cast<const f4*>(pa + ((it + r) & 7) * 256);
                const f4 nb = *reinterpret_cast<const f4*>(pb + ((it + r) & 7) * 256);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[(4 * r + i) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[i], b[i], acc[(4 * r + i) % NACC], 0, 0, 0);
                a[0] = na.x; a[1] = na.y; a[2] = na.z; a[3] = na.w;
                b[0] = nb.x; b[1] = nb.y; b[2] = nb.z; b[3] = nb.w;
            } else if (OP == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[(4 * r + i) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[i], b[i], acc[(4 * r + i) % NACC], 0, 0, 0);
            } else if (OP == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[(4 * r + i) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[i], b[i], acc[(4 * r + i) % NACC], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[(4 * i + e) % 12] = __builtin_fmaf(b[e], a[(e + i) & 3], v[(4 * i + e) % 12]);
                }
            } else if (OP == 4) {   // the walk's pattern: one A operand for a run of instructions, accumulators in VGPRs
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[(4 * r + i) % NACC]) : "v"(a[r & 3]), "v"(b[i]));
            } else if (OP == 5) {   // the same with the accumulators in AGPRs
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(acc[(4 * r + i) % NACC]) : "v"(a[r & 3]), "v"(b[i]));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 12; ++e) v[e] = __builtin_fmaf(b[i], a[(e + i) & 3], v[e]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP, int NACC>
void run(const char* name, int waves, double target_ms, bool forever = false) {
    float* out;
    const int grid = 256 * waves;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 500;
    float ms = 0;
    for (int pass = 0; pass < (forever ? 1000000 : 2); ++pass) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP, NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.25f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (pass == 0) iters = (int)(iters * target_ms / ms) + 1;
    }
    const double mfma = OP == 3 ? 0 : (double)iters * 32 * waves, vfma = OP == 2 ? (double)iters * 128 * waves : (OP == 3 ? (double)iters * 8 * 48 * waves : 0);
    const double macs = mfma * 256 + vfma * 64;
    printf("%-9s acc %2d waves/SIMD %d: %7.2f ms", name, NACC, waves, ms);
    if (mfma > 0) printf("  %.3f ns per mfma per SIMD", ms * 1e6 / mfma);
    if (vfma > 0) printf("  %.3f ns per v_fmac per SIMD", ms * 1e6 / vfma);
    printf("  = %6.1f TMAC/s\n", macs * 1024.0 / (ms * 1e-3) * 1e-12);
    hipFree(out);
}

int main(int argc, char** argv) {
    const double target = argc > 1 ? atof(argv[1]) : 20.0;
    if (argc > 2 && atoi(argv[2]) == 1) { run<1, 8>("mfma+lds", 2, target, true); return 0; }
    if (argc > 2 && atoi(argv[2]) == 2) { run<3, 8>("vfma", 2, target, true); return 0; }
    for (int w : {1, 2, 3, 4}) {
        run<0, 4>("mfma", w, target);
        run<0, 8>("mfma", w, target);
        run<1, 8>("mfma+lds", w, target);
        run<2, 8>("mix", w, target);
        run<3, 8>("vfma", w, target);
        run<4, 13>("mfma vgpr", w, target);
        run<5, 13>("mfma agpr", w, target);
    }
    return 0;
}
