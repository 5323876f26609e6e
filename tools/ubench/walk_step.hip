// The walk of the windowed FilteredNoise kernel (csrc/noise_win.hip: fir_win_core) in isolation: no design, no barriers,
// no global traffic inside the timed region -- how many SIMD cycles does one step (4 OPL multiply-adds fed by two
// ds_read_b128) cost a wavefront when 1, 2 or 3 workgroups share a CU?  clock64 (s_memtime) counts shader cycles, the
// events give wall time: their ratio is the clock the chip really holds under this load (round 4: 2.36 GHz with one
// wavefront per SIMD, 1.4 GHz with three -- the socket sits at its 1400 W limit, tools/power_probe.sh).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize \
//        -I ../../ddsp_piano_amd/csrc walk_step.hip ../../ddsp_piano_amd/csrc/error.cpp -o walk_step
// usage: walk_step [lds_pad_bytes] [reps] [CUs to fill]
#include "../../ddsp_piano_amd/csrc/noise_win.hip"

#ifndef WPE
#define WPE 3          // wavefronts per SIMD the walk kernel is compiled for (-DWPE=4: without the magnitude tile, four workgroups per CU)
#endif

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace ddspp;

template <int OPL, int BPF, int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
walk_kernel(WinGeom g, int reps, float* __restrict__ sink, long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    constexpr int D = WIN_D;
    float* Xs = lds_dyn;
    float* Gtop = Xs + (BPF + 1) * 4 * D;
    float* G = Gtop - g.gshift;
    for (int i = threadIdx.x; i < (BPF + 1) * 4 * D + D * g.gs - g.gshift; i += 256) lds_dyn[i] = 1e-3f * (float)(i & 63);
    __syncthreads();
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wibs = wave_uniform(wib);
    const int half = lane >> 5, fr = min(lane & 31, g.W - 1);
    float acc[OPL];
#pragma unroll
    for (int e = 0; e < OPL; ++e) acc[e] = 0.f;
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        const float *gl, *xl;
        win_lane<OPL, BPF>(g, G, Xs, fr, 2 * wib + half, gl, xl);
        const int qA = g.q_hi0 + (OPL / 4) * (2 * wibs);
        if (MODE == 0) fir_win_core<OPL, BPF>(gl, xl, qA + OPL / 4, qA - g.nsteps + 1, g.gs, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < OPL; ++e) s += acc[e];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wib] = t1 - t0;
}

int main(int argc, char** argv) {
    const size_t pad = argc > 1 ? (size_t)atoi(argv[1]) : 0;
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    const int T = 750, U = 96, N = T * U, Lw = 190;
    const int delay = (Lw - 1) / 2 - 1;
    WinGeom g;
    if (!win_tvfir_supported(N, T, Lw, delay, &g)) return 1;
    const size_t lds = win_lds_bytes(g, 0) + (WPE >= 4 ? 0 : 12800) + pad;      // + the fused kernel's magnitude tile, 32 rows of 96 + 4 floats (not with -DWPE=4)
    hipFuncSetAttribute(reinterpret_cast<const void*>(&walk_kernel<12, 24, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, walk_kernel<12, 24, 0>, 256, lds);
    const int ncu = argc > 3 ? atoi(argv[3]) : 256;          // fewer CUs enabled (CU mask): is the step time a power / clock effect?
    const int grid = ncu * nb;
    hipStream_t st = 0;
    if (ncu < 256) {
        std::vector<uint32_t> mask(8, 0u);
        for (int c = 0; c < ncu; ++c) { const int bit = (int)((long long)c * 256 / ncu); mask[bit / 32] |= 1u << (bit % 32); }
        if (hipExtStreamCreateWithCUMask(&st, 8, mask.data()) != hipSuccess) { printf("CU mask failed\n"); return 1; }
    }
    float* sink; long long* cyc;
    hipMalloc(&sink, (size_t)grid * 256 * 4);
    hipMalloc(&cyc, (size_t)grid * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, st);
        hipLaunchKernelGGL((walk_kernel<12, 24, 0>), dim3(grid), dim3(256), lds, st, g, reps, sink, cyc);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h((size_t)grid * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        const int steps = g.nsteps + 3;
        printf("lds %zu B, %d workgroups/CU x %d CUs: %.3f ms, %.0f clock64 ticks per walk (%d steps: %.1f per step), wall %.1f ns per step\n",
               lds, nb, ncu, ms, s / h.size() / reps, steps, s / h.size() / reps / steps, ms * 1e6 / reps / steps);
    }
    return 0;
}
