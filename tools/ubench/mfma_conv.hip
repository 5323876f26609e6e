// Layout check for the matrix-pipe FIR walk (noise_win.hip, fir_win_mfma): v_mfma_f32_4x4x1_16b_f32 as sixteen independent
// 4 x 4 outer products, D_b[i][j] += A_b[i] B_b[j] with A = four consecutive taps, B = four consecutive input samples of
// lane block b (lanes 4 b .. 4 b + 3); the anti-diagonal sums through quad_perm DPP.  One wavefront convolves sixteen
// independent (x_b, h_b) pairs and the host checks every output against a double-precision sum.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_conv.hip -o mfma_conv
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NX = 64, NH = 24, QB = 5;      // outputs 4 (QB - 1) = 16 per block: n = 4 c + m, c = 1 .. QB - 1

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// out[b][n] = sum_j x[b][j] h[b][n - j], n = 4 .. 4 QB - 1 (full overlap region chosen so that every index is in range)
__global__ void __launch_bounds__(64) k(const float* __restrict__ x, const float* __restrict__ h, float* __restrict__ out) {
    const int lane = threadIdx.x, b = lane >> 2, sub = lane & 3;
    const float* xb = x + b * NX;
    const float* hb = h + b * NH;
    f4 acc[QB];
#pragma unroll
    for (int c = 0; c < QB; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
    // D_c[i][jj] = sum_q h[4 (c - q) + i] x[4 q + jj]  ->  out[4 c + i + jj]
    for (int q = 0; q < NX / 4; ++q) {
        const float xv = xb[4 * q + sub];
#pragma unroll
        for (int c = 0; c < QB; ++c) {
            const int t = c - q;
            if (t >= 0 && 4 * t + 3 < NH) {                       // (wave-uniform)
                const float hv = hb[4 * t + sub];
                acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(hv, xv, acc[c], 0, 0, 0);
            }
        }
    }
    // quad c (outputs 4 c .. 4 c + 3) = the m < 4 part of D_c + the m >= 4 part of D_{c-1}
#pragma unroll
    for (int c = 1; c < QB; ++c) {
        float w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (sub + i >= 4) ? acc[c - 1][i] : acc[c][i];
        const float o = w[0] + dpp<0x93>(w[1]) + dpp<0x4E>(w[2]) + dpp<0x39>(w[3]);
        out[b * 4 * QB + 4 * c + sub] = o;
    }
}

int main() {
    std::vector<float> x(16 * NX), h(16 * NH), out(16 * 4 * QB, 0.f);
    for (auto& v : x) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *dx, *dh, *dout;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dh, h.size() * 4); hipMalloc(&dout, out.size() * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dout, 0, out.size() * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dh, dout);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int b = 0; b < 16; ++b)
        for (int n = 4; n < 4 * QB; ++n) {
            double ref = 0;
            for (int j = 0; j < NX; ++j) {
                const int t = n - j;
                if (t >= 0 && t < NH) ref += (double)x[b * NX + j] * h[b * NH + t];
            }
            worst = fmax(worst, fabs(ref - out[b * 4 * QB + n]));
        }
    printf("mfma_conv: worst |error| over 16 blocks x %d outputs = %.3g  (%s)\n", 4 * QB - 4, worst, worst < 1e-5 ? "layout OK" : "LAYOUT WRONG");
    return worst < 1e-5 ? 0 : 1;
}
