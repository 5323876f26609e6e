// Phase timeline of noise_win_fused_kernel (csrc/noise_win.hip) at the headline shape: the kernel is instantiated with
// TRACE = true and every wavefront of the first 64 workgroups records the shader clock at its phase boundaries.
//   marks: 0 unit start | 1 design done | 2 past barrier B | 3 walk done (+ output stores) | 4 magnitudes in LDS
//          | 5 past barrier C | 6 noise in LDS, next fetch issued
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I ../../ddsp_piano_amd/csrc \
//        noise_win_trace.hip ../../ddsp_piano_amd/csrc/error.cpp -o noise_win_trace
#include "../../ddsp_piano_amd/csrc/noise_win.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace ddspp;

int main(int argc, char** argv) {
    const int vq = argc > 1 ? atoi(argv[1]) : 8;
    const int units_per_wg = argc > 2 ? atoi(argv[2]) : 8;
    const int R = 1024, T = 750, U = 96, N = T * U, K = 96, Lw = 190, NJ = 48, n_voices = 16;
    const int delay = (Lw - 1) / 2 - 1;
    WinGeom g;
    if (!win_fused_supported(N, T, K, Lw, delay, &g)) return 1;
    const size_t lds = win_lds_bytes(g, K) + (argc > 3 ? (size_t)atoi(argv[3]) : 0);       // argv[3]: LDS pad -> fewer workgroups per CU
    printf("geometry: W %d gs %d padl %d nsteps %d lds %zu B\n", g.W, g.gs, g.padl, g.nsteps, lds);
    float *x, *mags, *CE, *CO, *we, *wo, *out, *out_last;
    int* ti;
    long long* trace;
    hipMalloc(&x, (size_t)R * N * 4);
    hipMalloc(&mags, (size_t)R * T * K * 4);
    hipMalloc(&out, (size_t)R * N * 4);
    hipMalloc(&out_last, (size_t)(R / n_voices) * N * 4);
    hipMalloc(&CE, 48 * 48 * 4); hipMalloc(&CO, 48 * 48 * 4); hipMalloc(&we, 48 * 16); hipMalloc(&wo, 48 * 16); hipMalloc(&ti, 48 * 16);
    const size_t trace_n = (size_t)WIN_TRACE_WGS * WIN_TRACE_UNITS * 4 * WIN_TRACE_MARKS;
    hipMalloc(&trace, trace_n * 8);
    hipMemset(trace, 0, trace_n * 8);
    {
        std::vector<float> h((size_t)R * N);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 32768.0f - 1.0f;
        hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> m((size_t)R * T * K);
        for (size_t i = 0; i < m.size(); ++i) m[i] = (float)((i * 40503u) >> 4 & 0xfff) / 1024.0f - 2.0f;
        hipMemcpy(mags, m.data(), m.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> c(48 * 48, 0.01f), w(48 * 4, 0.5f);
        std::vector<int> t(48 * 4);
        for (int j = 0; j < 48; ++j)
            for (int s = 0; s < 4; ++s) t[4 * j + s] = (4 * j + s < Lw) ? 4 * j + s : -1;
        hipMemcpy(CE, c.data(), c.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(CO, c.data(), c.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(we, w.data(), w.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(wo, w.data(), w.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(ti, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    }
    const ScaleFn sf{1, logf(10.0f), 2.0f, 1e-7f, 1.0f};
    const long long tasks = (long long)(R / vq) * g.wpr;
    int tpw = units_per_wg / vq;
    if (tpw < 1) tpw = 1;
    const dim3 grid((unsigned)((tasks + tpw - 1) / tpw)), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (argc > 4 && atoi(argv[4]) == 2) {       // argv[4] = 2: the matrix-pipe walk with the next unit's design riding
            hipLaunchKernelGGL((noise_win_fused_ride_kernel<48, 3, 24, 13, 93, 190, 1, 1, true>), grid, block, lds, 0, x, mags, CE, CO, ti,
                               we, wo, out, vq > 1 ? out_last : nullptr, R, N, T, NJ, g, -5.0f, sf, vq, n_voices, 0, tpw,
                               (int)((grid.x / WIN_TRACE_WGS) << 8), trace);
        } else if (argc > 4 && atoi(argv[4]) == 1)  // argv[4] = 1: the matrix-pipe walk
            hipLaunchKernelGGL((noise_win_fused_mw_kernel<48, 3, 12, 24, 13, 93, 190, 1, 1, true>), grid, block, lds, 0, x, mags, CE, CO, ti,
                               we, wo, out, vq > 1 ? out_last : nullptr, R, N, T, NJ, g, -5.0f, sf, vq, n_voices, 0, tpw,
                               (int)((grid.x / WIN_TRACE_WGS) << 8), trace);
        else
        hipLaunchKernelGGL((noise_win_fused_kernel<48, 3, 12, 24, true>), grid, block, lds, 0, x, mags, CE, CO, ti, we, wo, out,
                           vq > 1 ? out_last : nullptr, R, N, T, NJ, g, -5.0f, sf, vq, n_voices, 0, tpw, (int)((grid.x / WIN_TRACE_WGS) << 8), trace);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.3f ms (%u workgroups)\n", rep, ms, grid.x);
    }
    std::vector<long long> h(trace_n);
    hipMemcpy(h.data(), trace, trace_n * 8, hipMemcpyDeviceToHost);
    const char* names_v[7] = {"design", "barrierB", "walk", "store_m", "barrierC", "store_x", "(next)"};
    const char* names_r[7] = {"images", "store_x", "store_m", "fetches", "barrier", "walk+ride", "barrier2"};
    const char** names = (argc > 4 && atoi(argv[4]) == 2) ? names_r : names_v;
    // (ride kernel: images | store_x + store_m + fetches | barrier | walk + ride + outputs | - | barrier | -)
    // per-phase average over the traced workgroups / units, per wavefront index
    for (int w = 0; w < 4; ++w) {
        double acc[7] = {0};
        int cnt = 0;
        for (int b = 0; b < WIN_TRACE_WGS; ++b)
            for (int u = 1; u < WIN_TRACE_UNITS - 1; ++u) {
                const long long* m = &h[(((size_t)b * WIN_TRACE_UNITS + u) * 4 + w) * WIN_TRACE_MARKS];
                const long long* nx = &h[(((size_t)b * WIN_TRACE_UNITS + u + 1) * 4 + w) * WIN_TRACE_MARKS];
                if (m[6] == 0 || nx[0] == 0) continue;
                for (int k = 0; k < 6; ++k) acc[k] += (double)(m[k + 1] - m[k]);
                acc[6] += (double)(nx[0] - m[6]);
                ++cnt;
            }
        printf("wave %d (%d samples), cycles per unit:", w, cnt);
        double tot = 0;
        for (int k = 0; k < 7; ++k) {
            printf("  %s %.0f", names[k], acc[k] / (cnt ? cnt : 1));
            tot += acc[k] / (cnt ? cnt : 1);
        }
        printf("  | total %.0f\n", tot);
    }
    // when did the recorded workgroups (every (grid / 64)-th) run?  wall_clock64 ticks at 100 MHz
    {
        long long t0 = -1;
        for (int b = 0; b < WIN_TRACE_WGS; ++b) {
            const long long m = h[(((size_t)b * WIN_TRACE_UNITS) * 4) * WIN_TRACE_MARKS];
            if (m && (t0 < 0 || m < t0)) t0 = m;
        }
        printf("recorded workgroups: start .. end in us after the first start (one per line: index, start, end, per-unit us)\n");
        for (int b = 0; b < WIN_TRACE_WGS; b += 3) {
            const long long s0 = h[(((size_t)b * WIN_TRACE_UNITS) * 4) * WIN_TRACE_MARKS];
            const long long e0 = h[(((size_t)b * WIN_TRACE_UNITS + WIN_TRACE_UNITS - 1) * 4) * WIN_TRACE_MARKS + 6];
            printf("  wg %4d: %8.1f .. %8.1f  (%.2f us per unit)\n", b * (int)(grid.x / WIN_TRACE_WGS), (s0 - t0) * 0.01, (e0 - t0) * 0.01,
                   (e0 - s0) * 0.01 / WIN_TRACE_UNITS);
        }
    }
    // one workgroup in full
    for (int b : {0, 17}) {
        printf("workgroup %d, wave 0 / wave 3 marks relative to the unit start of wave 0:\n", b);
        for (int u = 0; u < WIN_TRACE_UNITS; ++u) {
            const long long* m0 = &h[(((size_t)b * WIN_TRACE_UNITS + u) * 4 + 0) * WIN_TRACE_MARKS];
            const long long* m3 = &h[(((size_t)b * WIN_TRACE_UNITS + u) * 4 + 3) * WIN_TRACE_MARKS];
            printf("  unit %d w0:", u);
            for (int k = 0; k < 7; ++k) printf(" %6lld", m0[k] - m0[0]);
            printf("   w3:");
            for (int k = 0; k < 7; ++k) printf(" %6lld", m3[k] - m0[0]);
            printf("\n");
        }
    }
    return 0;
}
