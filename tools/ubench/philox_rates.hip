// Round 6 probe: what the in-kernel noise draw (Philox4x32-10, noise_win.h) costs in issue slots, and whether forming a
// round's two 32 x 32 -> 64-bit products with one v_mad_u64_u32 each (the 64-bit product written as such) is cheaper than the
// v_mul_hi_u32 + v_mul_lo_u32 pair the compiler emits for __umulhi / `*`.  Same numbers either way (tested below).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/philox_rates tools/ubench/philox_rates.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <bool WIDE>
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0, lo0, hi1, lo1;
    if (WIDE) {
        const uint64_t p0 = (uint64_t)M0 * (uint64_t)c[0], p1 = (uint64_t)M1 * (uint64_t)c[2];
        hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
    } else {
        hi0 = __umulhi(M0, c[0]); lo0 = M0 * c[0]; hi1 = __umulhi(M1, c[2]); lo1 = M1 * c[2];
    }
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
}
template <bool WIDE, int ROUNDS>
__device__ __forceinline__ void philox(uint64_t seed, uint64_t ctr, uint32_t (&c)[4]) {
    c[0] = (uint32_t)ctr; c[1] = (uint32_t)(ctr >> 32); c[2] = 0u; c[3] = 0u;
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) philox_round<WIDE>(c, k);
}

// OP 0: mul_lo, 1: mul_hi, 2: mad_u64_u32 (64-bit product), 3: xor, 4: Philox pair form, 5: Philox wide form
template <int OP>
__global__ void __launch_bounds__(256) rate(uint32_t* out, int iters, uint32_t m) {
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 2654435761u + i;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (OP < 4) {
#pragma unroll
            for (int r = 0; r < 32; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (OP == 0) x[i] = x[i] * m;
                    if (OP == 1) x[i] = __umulhi(x[i], m) + 1u;
                    if (OP == 2) { const uint64_t p = (uint64_t)x[i] * (uint64_t)m; x[i] = (uint32_t)(p >> 32) ^ (uint32_t)p; }
                    if (OP == 3) x[i] = x[i] ^ m;
                }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t c[4];
                philox<OP == 5, 10>(m, ((uint64_t)it << 20) + x[i] + i, c);
                acc ^= c[0] ^ c[1] ^ c[2] ^ c[3];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void same(uint32_t* diff, uint64_t seed) {
    uint32_t a[4], b[4];
    const uint64_t ctr = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    philox<false, 10>(seed, ctr, a);
    philox<true, 10>(seed, ctr, b);
    if (a[0] != b[0] || a[1] != b[1] || a[2] != b[2] || a[3] != b[3]) atomicAdd(diff, 1u);
}

template <int OP>
void run(const char* name, double ops_per_iter, int waves) {
    uint32_t* out;
    hipMalloc(&out, 256 * 16 * 256 * 4);
    const int iters = OP < 4 ? 2000 : 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<OP>, dim3(256 * waves), dim3(256), 0, 0, out, 10, 0xD2511F53u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<OP>, dim3(256 * waves), dim3(256), 0, 0, out, iters, 0xD2511F53u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / ((double)iters * ops_per_iter * waves);
    printf("%-44s waves/SIMD=%d  %8.3f ms  %7.2f ns per %s per SIMD\n", name, waves, ms, per, OP < 4 ? "wave64 instruction" : "Philox4x32-10 counter (4 numbers per lane)");
    hipFree(out);
}

int main() {
    uint32_t* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipLaunchKernelGGL(same, dim3(4096), dim3(256), 0, 0, d, 0x123456789abcdefull);
    uint32_t h = 1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("pair form vs 64-bit-product form on 1 048 576 counters: %u differ\n", h);
    for (int w : {2, 4}) {
        run<0>("v_mul_lo_u32", 256, w);
        run<1>("v_mul_hi_u32 (+ v_add)", 256, w);
        run<2>("v_mad_u64_u32 (+ v_xor)", 256, w);
        run<3>("v_xor_b32", 256, w);
        run<4>("Philox, v_mul_hi + v_mul_lo pairs", 4, w);
        run<5>("Philox, 64-bit products (v_mad_u64_u32)", 4, w);
    }
    return 0;
}
