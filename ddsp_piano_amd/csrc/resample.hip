// Frame -> sample control upsamplers (stand-alone operators).
//
// Replaces ddsp.core.resample(method='linear') and ddsp.core.upsample_with_windows as called from
// ddsp_piano/modules/inharm_synth.py:117-119.  Both are pure streaming kernels: x[R, T, C] is tiny
// and L2 resident, y[R, N, C] is written once with 16-byte stores (C innermost, as the reference
// lays the envelopes out).  The index/weight tables are built by the Python host
// (ddsp_piano_amd/core.py) with the legacy-bilinear float32 arithmetic of the TF kernel.
#include "ddspp_common.h"

namespace ddspp {

// y[r, n, c] = x[r, lo[n], c] + (x[r, hi[n], c] - x[r, lo[n], c]) * w[n]
template <int VEC>
__global__ void __launch_bounds__(256) resample_linear_kernel(const float* __restrict__ x,
                                                            const int* __restrict__ lo,
                                                            const int* __restrict__ hi,
                                                            const float* __restrict__ w,
                                                            float* __restrict__ y, int R, int T, int C,
                                                            int N) {
    const int cv = C / VEC;
    const size_t total = (size_t)R * N * cv;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int c = (int)(g % cv) * VEC;
        const size_t rn = g / cv;
        const int n = (int)(rn % N);
        const int r = (int)(rn / N);
        const float wn = w[n];
        const float* xl = x + ((size_t)r * T + lo[n]) * C + c;
        const float* xh = x + ((size_t)r * T + hi[n]) * C + c;
        float* yo = y + (rn * C + c);
        if (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4*>(xl);
            const float4 b = *reinterpret_cast<const float4*>(xh);
            float4 o;
            o.x = a.x + (b.x - a.x) * wn;
            o.y = a.y + (b.y - a.y) * wn;
            o.z = a.z + (b.z - a.z) * wn;
            o.w = a.w + (b.w - a.w) * wn;
            *reinterpret_cast<float4*>(yo) = o;
        } else {
            yo[0] = xl[0] + (xh[0] - xl[0]) * wn;
        }
    }
}

// y[r, t*U + j, c] = x[r, t, c] * win[U + j] + x[r, min(t + 1, T - 1), c] * win[j]
// (= overlap_and_add of Hann-windowed frames with the appended end point, trimmed by one hop)
template <int VEC>
__global__ void __launch_bounds__(256) resample_window_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ win,
                                                            float* __restrict__ y, int R, int T, int C,
                                                            int U) {
    const int cv = C / VEC;
    const int N = T * U;
    const size_t total = (size_t)R * N * cv;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int c = (int)(g % cv) * VEC;
        const size_t rn = g / cv;
        const int n = (int)(rn % N);
        const int r = (int)(rn / N);
        const int t = n / U, j = n - t * U;
        const int t1 = min(t + 1, T - 1);
        const float w0 = win[U + j], w1 = win[j];
        const float* xa = x + ((size_t)r * T + t) * C + c;
        const float* xb = x + ((size_t)r * T + t1) * C + c;
        float* yo = y + (rn * C + c);
        if (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4*>(xa);
            const float4 b = *reinterpret_cast<const float4*>(xb);
            float4 o;
            o.x = a.x * w0 + b.x * w1;
            o.y = a.y * w0 + b.y * w1;
            o.z = a.z * w0 + b.z * w1;
            o.w = a.w * w0 + b.w * w1;
            *reinterpret_cast<float4*>(yo) = o;
        } else {
            yo[0] = xa[0] * w0 + xb[0] * w1;
        }
    }
}

// amplitude_envelopes[r, n, c] *= |decays[r, t, c]| ** (decay_time[r, t] * U + n % U),  t = n / U
// (surrogate_synth.py:76-95: tf.repeat of the frame values, sample counter added, tf.math.pow)
__global__ void __launch_bounds__(256) decay_envelope_kernel(float* __restrict__ env,
                                                           const float* __restrict__ decays,
                                                           const float* __restrict__ decay_time, int R, int T,
                                                           int C, int U) {
    const int N = T * U;
    const size_t total = (size_t)R * N * C;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int c = (int)(g % C);
        const size_t rn = g / C;
        const int n = (int)(rn % N), r = (int)(rn / N);
        const int t = n / U, j = n - t * U;
        const float tt = decay_time[(size_t)r * T + t] * (float)U + (float)j;
        env[g] = env[g] * powf(fabsf(decays[((size_t)r * T + t) * C + c]), tt);
    }
}

static unsigned stream_grid(size_t total) {
    size_t blocks = (total + 255) / 256;
    const size_t cap = 256 * 16;     // 256 CUs x 16 blocks, grid-stride beyond that
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace ddspp

using namespace ddspp;

extern "C" {

// ddsp.core.resample(x, n_timesteps, method='linear')  -- call site inharm_synth.py:117
int ddspp_resample_linear(const float* x, const int* lo, const int* hi, const float* w, float* y, int R,
                          int T, int C, int N, hipStream_t stream) {
    DDSPP_REQUIRE(x && lo && hi && w && y, "resample_linear: null buffer");
    DDSPP_REQUIRE(R > 0 && T > 0 && C > 0 && N > 0, "resample_linear: bad dims");
    const bool vec = (C % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    const size_t total = (size_t)R * N * (vec ? C / 4 : C);
    if (vec)
        hipLaunchKernelGGL(resample_linear_kernel<4>, dim3(stream_grid(total)), dim3(256), 0, stream, x,
                           lo, hi, w, y, R, T, C, N);
    else
        hipLaunchKernelGGL(resample_linear_kernel<1>, dim3(stream_grid(total)), dim3(256), 0, stream, x,
                           lo, hi, w, y, R, T, C, N);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// ddsp.core.resample(x, n_timesteps, method='window') = upsample_with_windows(add_endpoint=True)
// -- call site inharm_synth.py:118-119
int ddspp_resample_window(const float* x, const float* window, float* y, int R, int T, int C, int U,
                          hipStream_t stream) {
    DDSPP_REQUIRE(x && window && y, "resample_window: null buffer");
    DDSPP_REQUIRE(R > 0 && T > 0 && C > 0 && U > 1, "resample_window: bad dims");
    const bool vec = (C % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    const size_t total = (size_t)R * T * U * (vec ? C / 4 : C);
    if (vec)
        hipLaunchKernelGGL(resample_window_kernel<4>, dim3(stream_grid(total)), dim3(256), 0, stream, x,
                           window, y, R, T, C, U);
    else
        hipLaunchKernelGGL(resample_window_kernel<1>, dim3(stream_grid(total)), dim3(256), 0, stream, x,
                           window, y, R, T, C, U);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// surrogate_harmonic_synthesis' decay envelope (surrogate_synth.py:76-95), applied in place to the
// sample-rate amplitude envelopes [R, T*U, C]; decays [R, T, C], decay_time [R, T].
int ddspp_decay_envelope(float* amplitude_envelopes, const float* decays, const float* decay_time, int R, int T,
                         int C, int U, hipStream_t stream) {
    DDSPP_REQUIRE(amplitude_envelopes && decays && decay_time, "decay_envelope: null buffer");
    DDSPP_REQUIRE(R > 0 && T > 0 && C > 0 && U > 0, "decay_envelope: bad dims");
    const size_t total = (size_t)R * T * U * C;
    hipLaunchKernelGGL(decay_envelope_kernel, dim3(stream_grid(total)), dim3(256), 0, stream, amplitude_envelopes,
                       decays, decay_time, R, T, C, U);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

}  // extern "C"
