// Frame -> sample control upsamplers (stand-alone operators).
//
// Replaces ddsp.core.resample(method='linear') and ddsp.core.upsample_with_windows as called from
// ddsp_piano/modules/inharm_synth.py:117-119.  Both are pure streaming kernels: x[R, T, C] is tiny
// and L2 resident, y[R, N, C] is written once with 16-byte stores (C innermost, as the reference
// lays the envelopes out).  The index/weight tables are built by the Python host
// (ddsp_piano_amd/core.py) with the legacy-bilinear float32 arithmetic of the TF kernel.
#include "ddspp_common.h"

namespace ddspp {

// Both upsamplers are pure write streams ([R, N, C] float32 out of an L2-resident [R, T, C]): one workgroup = one row x
// one tile of RS_TILE samples, a thread owns a 16-byte column piece and steps through the tile's samples with ADDITIONS only
// (round 5's flat grid-stride loop spent two 64-bit divisions and two 64-bit remainders per element and ran at 0.41-0.44 of
// the HBM peak: VALU bound), four outputs in flight per thread, non-temporal 16-byte stores (the envelopes are consumed by
// another kernel much later or never from cache: nothing of 2.4 GB per voice should displace x in L2).
constexpr int RS_NT_DEFAULT = 1;  // non-temporal stores unless DDSPP_RESAMPLE_NT=0 (A/B: profiles/r06_ubench.txt)
constexpr int RS_TILE = 768;      // samples per workgroup: a multiple of 256 / gcd(256, C / VEC) for every C / VEC <= 256 that matters

typedef float rs_f4 __attribute__((ext_vector_type(4)));

template <bool NT, class V>
__device__ __forceinline__ void rs_store(V v, V* p) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// thread -> (column piece c, first sample nl of the tile), and its step: 256 consecutive pieces later
struct RsWalk {
    int c, nl, dc, dn, cv;
    __device__ __forceinline__ RsWalk(int cv_) : cv(cv_) {
        c = (int)threadIdx.x % cv;
        nl = (int)threadIdx.x / cv;
        dc = 256 % cv;
        dn = 256 / cv;
    }
    __device__ __forceinline__ void step() {
        c += dc;
        nl += dn;
        if (c >= cv) {
            c -= cv;
            ++nl;
        }
    }
};

// y[r, n, c] = x[r, lo[n], c] + (x[r, hi[n], c] - x[r, lo[n], c]) * w[n]
template <int VEC, bool NT>
__global__ void __launch_bounds__(256) resample_linear_kernel(const float* __restrict__ x,
                                                            const int* __restrict__ lo,
                                                            const int* __restrict__ hi,
                                                            const float* __restrict__ w,
                                                            float* __restrict__ y, int T, int C, int N, int tiles) {
    const int r = blockIdx.x / tiles;
    const int n0 = (blockIdx.x - r * tiles) * RS_TILE;
    const int cnt = min(RS_TILE, N - n0);
    const float* xr = x + (size_t)r * T * C;
    float* yr = y + ((size_t)r * N + n0) * C;
    lo += n0; hi += n0; w += n0;
    RsWalk k(C / VEC);
    const int iters = (cnt * k.cv + 255) / 256;
    for (int it = 0; it < iters; it += 4) {
        // (it, piece) -> loads first, then the arithmetic and the stores: four 16-byte stores in flight per thread
        int nl[4], c[4];
        bool inside[4];
        float wn[4];
        rs_f4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            inside[u] = k.nl < cnt;
            nl[u] = min(k.nl, cnt - 1);          // past the tile: a valid address, the store below is skipped
            c[u] = k.c * VEC;
            k.step();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = nl[u];
            wn[u] = w[n];
            const float* xl = xr + (size_t)lo[n] * C + c[u];
            const float* xh = xr + (size_t)hi[n] * C + c[u];
            if (VEC == 4) {
                a[u] = *reinterpret_cast<const rs_f4*>(xl);
                b[u] = *reinterpret_cast<const rs_f4*>(xh);
            } else {
                a[u].x = xl[0];
                b[u].x = xh[0];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!inside[u]) continue;
            float* yo = yr + (size_t)nl[u] * C + c[u];
            if (VEC == 4) {
                rs_f4 o;
                o.x = a[u].x + (b[u].x - a[u].x) * wn[u];
                o.y = a[u].y + (b[u].y - a[u].y) * wn[u];
                o.z = a[u].z + (b[u].z - a[u].z) * wn[u];
                o.w = a[u].w + (b[u].w - a[u].w) * wn[u];
                rs_store<NT>(o, reinterpret_cast<rs_f4*>(yo));
            } else {
                rs_store<NT>(a[u].x + (b[u].x - a[u].x) * wn[u], yo);
            }
        }
    }
}

// y[r, t*U + j, c] = x[r, t, c] * win[U + j] + x[r, min(t + 1, T - 1), c] * win[j]
// (= overlap_and_add of Hann-windowed frames with the appended end point, trimmed by one hop)
template <int VEC, bool NT>
__global__ void __launch_bounds__(256) resample_window_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ win,
                                                            float* __restrict__ y, int T, int C, int U, int tiles) {
    const int N = T * U;
    const int r = blockIdx.x / tiles;
    const int n0 = (blockIdx.x - r * tiles) * RS_TILE;
    const int cnt = min(RS_TILE, N - n0);
    const float* xr = x + (size_t)r * T * C;
    float* yr = y + ((size_t)r * N + n0) * C;
    RsWalk k(C / VEC);
    const int iters = (cnt * k.cv + 255) / 256;
    // (frame, sample of the frame) of the thread's current output, advanced with the walk: no division per element
    int t = (n0 + k.nl) / U, j = (n0 + k.nl) - t * U;
    for (int it = 0; it < iters; it += 4) {
        int nl[4], c[4], tt[4], jj[4];
        rs_f4 a[4], b[4];
        float w0[4], w1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            nl[u] = k.nl < cnt ? k.nl : -1;
            c[u] = k.c * VEC;
            tt[u] = min(t, T - 1);
            jj[u] = j;
            const int before = k.nl;
            k.step();
            j += k.nl - before;
            while (j >= U) {
                j -= U;
                ++t;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t1 = min(tt[u] + 1, T - 1);
            w0[u] = win[U + jj[u]];
            w1[u] = win[jj[u]];
            const float* xa = xr + (size_t)tt[u] * C + c[u];
            const float* xb = xr + (size_t)t1 * C + c[u];
            if (VEC == 4) {
                a[u] = *reinterpret_cast<const rs_f4*>(xa);
                b[u] = *reinterpret_cast<const rs_f4*>(xb);
            } else {
                a[u].x = xa[0];
                b[u].x = xb[0];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (nl[u] < 0) continue;
            float* yo = yr + (size_t)nl[u] * C + c[u];
            if (VEC == 4) {
                rs_f4 o;
                o.x = a[u].x * w0[u] + b[u].x * w1[u];
                o.y = a[u].y * w0[u] + b[u].y * w1[u];
                o.z = a[u].z * w0[u] + b[u].z * w1[u];
                o.w = a[u].w * w0[u] + b[u].w * w1[u];
                rs_store<NT>(o, reinterpret_cast<rs_f4*>(yo));
            } else {
                rs_store<NT>(a[u].x * w0[u] + b[u].x * w1[u], yo);
            }
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
    const int tiles = (N + RS_TILE - 1) / RS_TILE;
    DDSPP_REQUIRE((long long)R * tiles < (1ll << 31), "resample_linear: R x N too large for one launch");
    const dim3 grid((unsigned)(R * tiles));
    const bool nt = ddspp_option_literal("DDSPP_RESAMPLE_NT", RS_NT_DEFAULT) != 0;
    if (vec && nt) hipLaunchKernelGGL((resample_linear_kernel<4, true>), grid, dim3(256), 0, stream, x, lo, hi, w, y, T, C, N, tiles);
    else if (vec) hipLaunchKernelGGL((resample_linear_kernel<4, false>), grid, dim3(256), 0, stream, x, lo, hi, w, y, T, C, N, tiles);
    else if (nt) hipLaunchKernelGGL((resample_linear_kernel<1, true>), grid, dim3(256), 0, stream, x, lo, hi, w, y, T, C, N, tiles);
    else hipLaunchKernelGGL((resample_linear_kernel<1, false>), grid, dim3(256), 0, stream, x, lo, hi, w, y, T, C, N, tiles);
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
    DDSPP_REQUIRE((long long)T * U < (1ll << 31), "resample_window: n_samples does not fit an int");
    const int tiles = (T * U + RS_TILE - 1) / RS_TILE;
    DDSPP_REQUIRE((long long)R * tiles < (1ll << 31), "resample_window: R x N too large for one launch");
    const dim3 grid((unsigned)(R * tiles));
    const bool nt = ddspp_option_literal("DDSPP_RESAMPLE_NT", RS_NT_DEFAULT) != 0;
    if (vec && nt) hipLaunchKernelGGL((resample_window_kernel<4, true>), grid, dim3(256), 0, stream, x, window, y, T, C, U, tiles);
    else if (vec) hipLaunchKernelGGL((resample_window_kernel<4, false>), grid, dim3(256), 0, stream, x, window, y, T, C, U, tiles);
    else if (nt) hipLaunchKernelGGL((resample_window_kernel<1, true>), grid, dim3(256), 0, stream, x, window, y, T, C, U, tiles);
    else hipLaunchKernelGGL((resample_window_kernel<1, false>), grid, dim3(256), 0, stream, x, window, y, T, C, U, tiles);
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
