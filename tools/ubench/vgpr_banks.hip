// Micro-benchmark: does the issue cost of a wave64 VALU instruction on gfx950 depend on which VGPRs its operands sit in
// (register-file bank conflicts, bank = register index mod 4)?  Hand-placed registers, 8 independent destinations.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define R8(op, d, a, b, c)                                      \
    op " v" #d ", v" #a ", v" #b ", v" #c "\n"
template <int CASE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    // v40..v47: destinations; sources initialised once
    asm volatile(
        "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n"
        "v_mov_b32 v24, 0.5\n v_mov_b32 v25, 0.5\n v_mov_b32 v26, 0.5\n v_mov_b32 v27, 0.5\n"
        "v_mov_b32 v28, 0.5\n v_mov_b32 v29, 0.5\n v_mov_b32 v30, 0.5\n v_mov_b32 v31, 0.5\n"
        ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (CASE == 0)   // fma: three sources in three different banks (20:0, 25:1, 30:2)
                asm volatile(R8("v_fma_f32", 40, 20, 25, 30) R8("v_fma_f32", 41, 20, 25, 30) R8("v_fma_f32", 42, 20, 25, 30)
                             R8("v_fma_f32", 43, 20, 25, 30) R8("v_fma_f32", 44, 20, 25, 30) R8("v_fma_f32", 45, 20, 25, 30)
                             R8("v_fma_f32", 46, 20, 25, 30) R8("v_fma_f32", 47, 20, 25, 30)
                             ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
            if (CASE == 1)   // fma: three sources in ONE bank (20, 24, 28: bank 0)
                asm volatile(R8("v_fma_f32", 40, 20, 24, 28) R8("v_fma_f32", 41, 20, 24, 28) R8("v_fma_f32", 42, 20, 24, 28)
                             R8("v_fma_f32", 43, 20, 24, 28) R8("v_fma_f32", 44, 20, 24, 28) R8("v_fma_f32", 45, 20, 24, 28)
                             R8("v_fma_f32", 46, 20, 24, 28) R8("v_fma_f32", 47, 20, 24, 28)
                             ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
            if (CASE == 2)   // fma: two sources in one bank (20, 24), third elsewhere (29)
                asm volatile(R8("v_fma_f32", 40, 20, 24, 29) R8("v_fma_f32", 41, 20, 24, 29) R8("v_fma_f32", 42, 20, 24, 29)
                             R8("v_fma_f32", 43, 20, 24, 29) R8("v_fma_f32", 44, 20, 24, 29) R8("v_fma_f32", 45, 20, 24, 29)
                             R8("v_fma_f32", 46, 20, 24, 29) R8("v_fma_f32", 47, 20, 24, 29)
                             ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
            if (CASE == 3)   // add: two sources in different banks
                asm volatile("v_add_f32 v40, v20, v25\n v_add_f32 v41, v20, v25\n v_add_f32 v42, v20, v25\n v_add_f32 v43, v20, v25\n"
                             "v_add_f32 v44, v20, v25\n v_add_f32 v45, v20, v25\n v_add_f32 v46, v20, v25\n v_add_f32 v47, v20, v25\n"
                             ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
            if (CASE == 4)   // add: two sources in the same bank
                asm volatile("v_add_f32 v40, v20, v24\n v_add_f32 v41, v20, v24\n v_add_f32 v42, v20, v24\n v_add_f32 v43, v20, v24\n"
                             "v_add_f32 v44, v20, v24\n v_add_f32 v45, v20, v24\n v_add_f32 v46, v20, v24\n v_add_f32 v47, v20, v24\n"
                             ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
            if (CASE == 5)   // fmac with a 32-bit literal (the 2 pi reduction): v = v + lit * v
                asm volatile("v_fmac_f32 v40, 0xc0c90fdb, v25\n v_fmac_f32 v41, 0xc0c90fdb, v25\n v_fmac_f32 v42, 0xc0c90fdb, v25\n v_fmac_f32 v43, 0xc0c90fdb, v25\n"
                             "v_fmac_f32 v44, 0xc0c90fdb, v25\n v_fmac_f32 v45, 0xc0c90fdb, v25\n v_fmac_f32 v46, 0xc0c90fdb, v25\n v_fmac_f32 v47, 0xc0c90fdb, v25\n"
                             ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
            if (CASE == 6)   // add: a source is the destination of the instruction two before (the phase chain, two chains)
                asm volatile("v_add_f32 v40, v40, v25\n v_add_f32 v41, v41, v25\n v_add_f32 v40, v40, v25\n v_add_f32 v41, v41, v25\n"
                             "v_add_f32 v40, v40, v25\n v_add_f32 v41, v41, v25\n v_add_f32 v40, v40, v25\n v_add_f32 v41, v41, v25\n"
                             ::: "v40", "v41");
            if (CASE == 7)   // fma with an SGPR operand (VOP3, 8 bytes)
                asm volatile("v_fma_f32 v40, s20, v25, v30\n v_fma_f32 v41, s20, v25, v30\n v_fma_f32 v42, s20, v25, v30\n v_fma_f32 v43, s20, v25, v30\n"
                             "v_fma_f32 v44, s20, v25, v30\n v_fma_f32 v45, s20, v25, v30\n v_fma_f32 v46, s20, v25, v30\n v_fma_f32 v47, s20, v25, v30\n"
                             ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        }
    }
    float r;
    asm volatile("v_mov_b32 %0, v40" : "=v"(r));
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int CASE>
void run(const char* name, int blocks_per_cu) {
    float* out;
    hipMalloc(&out, 256 * 16 * 256 * 4);
    int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CASE>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CASE>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 16 * 8 * blocks_per_cu;
    printf("%-58s waves/SIMD=%d  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, blocks_per_cu, ms * 1e6 / n * 2.4);
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 4}) {
        run<0>("v_fma  sources in 3 different banks", w);
        run<1>("v_fma  3 sources in one bank", w);
        run<2>("v_fma  2 sources in one bank", w);
        run<3>("v_add  sources in 2 banks", w);
        run<4>("v_add  2 sources in one bank", w);
        run<5>("v_fmac with 32-bit literal", w);
        run<6>("v_add  2 dependent chains (dst = src of 2 before)", w);
        run<7>("v_fma  with an SGPR source", w);
    }
    return 0;
}
