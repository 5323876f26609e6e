// Micro-benchmark: issue interval of DEPENDENT wave64 VALU instructions (one wavefront per SIMD), as a function of
// the number of independent chains per wavefront.  Explains why compiler-serialised per-sample chains
// (add -> mul -> rndne -> fma -> mul -> cos -> fma through two temporaries) run far below the 2.3 cycle issue rate.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP 64
template <int CHAINS, int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (OP == 0) x[i] = __builtin_fmaf(x[i], a, b);
                if (OP == 1) x[i] = __builtin_amdgcn_cosf(x[i]);
                if (OP == 2) x[i] = __builtin_rintf(x[i]) + b;       // two dependent ops
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS, int OP>
void run(const char* name, int blocks_per_cu, int ops) {
    float* out;
    (void)hipMalloc(&out, 256 * 16 * 256 * 4);
    int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CHAINS, OP>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, OP>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double instrs_per_simd = (double)iters * REP * CHAINS * ops * blocks_per_cu;
    printf("%-10s chains/wave=%d waves/SIMD=%d  %.2f cyc per wave-instr per SIMD (2.4 GHz)\n", name, CHAINS, blocks_per_cu,
           ms * 1e6 / instrs_per_simd * 2.4);
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<1, 0>("fma", w, 1); run<2, 0>("fma", w, 1); run<4, 0>("fma", w, 1); run<8, 0>("fma", w, 1);
        run<1, 1>("cos", w, 1); run<2, 1>("cos", w, 1); run<4, 1>("cos", w, 1);
        run<1, 2>("rndne+add", w, 2); run<4, 2>("rndne+add", w, 2);
    }
    return 0;
}
