// Time-varying FilteredNoise FIR for gfx950.
//
// Replaces ddsp.core.frequency_filter -> frequency_impulse_response -> fft_convolve as reached from
// DynamicSizeFilteredNoise.get_signal (ddsp_piano/modules/filtered_noise_synth.py:27-42):
//   ir_t   = window * irfft(magnitudes[t])  (zero phase -> causal linear phase, Lw taps)
//   z[m]   = sum_j noise[j] * ir_{j / U}[m - j]           (each U-sample block filtered by its own
//                                                          frame's FIR, tails overlap-added)
//   out[n] = z[n + delay],  delay = (Lw - 1) // 2 - 1     (crop_and_compensate_delay, 'same')
// The reference evaluates the block convolutions with 512-point FFTs; for Lw <= 254 taps and
// U <= 192 samples the direct form is cheaper on CDNA4 than three FFTs per frame plus their HBM
// round trips, and it equals the FFT result to float32 round-off.
//
// Kernels
//   fir_from_magnitudes_kernel : ir[r, t, :] = magnitudes[r, t, :] @ M, M = the (windowed,
//       shifted) inverse real DFT matrix [K, Lw] built by the host in float64.  Lane = tap, the
//       tap's column of M lives in registers, the frame's magnitudes arrive through scalar loads.
//   tv_fir_kernel : gather form, one wavefront per 512 output samples.  Lane a owns 4 consecutive
//       outputs; noise samples are wave-uniform scalars (s_load), the frame FIRs are staged in LDS
//       and read as aligned 16-byte blocks that slide by one block per 4 input samples
//       (16 FMAs per ds_read_b128).
//   tv_fir_generic_kernel : thread-per-output fallback for shapes the tiled kernel does not take.
//   uniform_noise_kernel : Philox4x32-10 counter based U(-1, 1) noise (the reference draws an
//       unseeded tf.random.uniform; parity is defined with the noise tensor supplied).
#include "ddspp_common.h"

namespace ddspp {

// ------------------------------------------------------------------------------------------------
// impulse responses from magnitudes
// ------------------------------------------------------------------------------------------------
// Lane j computes the unique tap uniq[j] (and writes its mirror image when the FIR is symmetric:
// the zero-phase -> linear-phase construction makes ir[c + m] == ir[c - m]).  The K magnitudes of a
// frame are wave-uniform: they arrive through scalar loads, 16 at a time, and feed the FMAs as SGPR
// operands; the tap's column of M stays in K VGPRs for the whole run of frames.
template <int K, int THREADS>
__global__ void __launch_bounds__(THREADS) fir_from_magnitudes_kernel(const float* __restrict__ mags,
                                                                    const float* __restrict__ M,
                                                                    const int* __restrict__ uniq,
                                                                    const int* __restrict__ mirror,
                                                                    float* __restrict__ ir, int frames,
                                                                    int Lw, int n_uniq,
                                                                    int frames_per_block) {
    const int j = threadIdx.x;
    const bool active = j < n_uniq;
    const int tap = active ? uniq[j] : 0;
    const int mir = active ? mirror[j] : -1;
    float col[K];
#pragma unroll
    for (int k = 0; k < K; ++k) col[k] = active ? M[(size_t)k * Lw + tap] : 0.0f;
    const int f0 = blockIdx.x * frames_per_block;
    const int f1 = min(f0 + frames_per_block, frames);
    for (int f = f0; f < f1; ++f) {
        const float* __restrict__ mg = mags + (size_t)f * K;       // wave-uniform address
        float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
        for (int k = 0; k < K; k += 2) {
            acc0 = __builtin_fmaf(mg[k], col[k], acc0);
            acc1 = __builtin_fmaf(mg[k + 1], col[k + 1], acc1);
        }
        const float acc = acc0 + acc1;
        if (active) {
            ir[(size_t)f * Lw + tap] = acc;
            if (mir >= 0) ir[(size_t)f * Lw + mir] = acc;
        }
    }
}

// Even/odd split of the inverse real DFT (full-window case, Lh = 2 (K - 1), half = K - 1):
//   cos(2 pi (half - j) k / Lh) = (-1)^k cos(2 pi j k / Lh)
//   E[j] = sum_{k even} c_k m_k cos(.)/Lh,  O[j] = sum_{k odd} ...,   z[j] = E + O,  z[half - j] = E - O
// so lane j (j <= half / 2) yields up to four taps of the symmetric FIR from K FMAs -- a quarter of the
// dense product.  The frame's magnitudes are staged in LDS (coalesced) and read back as wave-uniform
// 16-byte blocks; the lane's two table columns stay in K registers.
template <int KH>   // KH = K / 2 (K even)
__global__ void __launch_bounds__(256) fir_eo_kernel(const float* __restrict__ mags,     // [frames, 2 KH]
                                                   const float* __restrict__ CE,       // [KH, NJ]
                                                   const float* __restrict__ CO,       // [KH, NJ]
                                                   const int* __restrict__ tap_idx,    // [NJ, 4]
                                                   const float* __restrict__ tap_we,   // [NJ, 4]
                                                   const float* __restrict__ tap_wo,   // [NJ, 4]
                                                   float* __restrict__ ir, int frames, int Lw, int NJ,
                                                   int frames_per_block) {
    extern __shared__ __attribute__((aligned(16))) float mtile[];       // [frames_per_block][2 KH]
    constexpr int K = 2 * KH;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int j = min(lane, NJ - 1);
    const bool active = lane < NJ;
    float ce[KH], co[KH];
#pragma unroll
    for (int q = 0; q < KH; ++q) {
        ce[q] = CE[q * NJ + j];
        co[q] = CO[q * NJ + j];
    }
    int tidx[4];
    float twe[4], two[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        tidx[s] = active ? tap_idx[j * 4 + s] : -1;
        twe[s] = tap_we[j * 4 + s];
        two[s] = tap_wo[j * 4 + s];
    }
    const int f0 = blockIdx.x * frames_per_block;
    const int nf = min(frames_per_block, frames - f0);
    {   // coalesced copy of the tile
        const float4* src = reinterpret_cast<const float4*>(mags + (size_t)f0 * K);
        float4* dst = reinterpret_cast<float4*>(mtile);
        const int n4 = nf * (K / 4);
        for (int i = threadIdx.x; i < n4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    for (int f = wib; f < nf; f += 4) {
        const float4* mg = reinterpret_cast<const float4*>(mtile + f * K);   // wave-uniform address
        float e0 = 0.f, e1 = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int q = 0; q < KH; q += 2) {
            const float4 m = mg[q / 2];                  // magnitudes 2q, 2q+1, 2q+2, 2q+3
            e0 = __builtin_fmaf(m.x, ce[q], e0);
            o0 = __builtin_fmaf(m.y, co[q], o0);
            e1 = __builtin_fmaf(m.z, ce[q + 1], e1);
            o1 = __builtin_fmaf(m.w, co[q + 1], o1);
        }
        const float E = e0 + e1, O = o0 + o1;
        float* dst = ir + (size_t)(f0 + f) * Lw;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (tidx[s] >= 0) dst[tidx[s]] = __builtin_fmaf(two[s], O, twe[s] * E);
    }
}

__global__ void __launch_bounds__(256) fir_from_magnitudes_generic_kernel(const float* __restrict__ mags,
                                                                        const float* __restrict__ M,
                                                                        float* __restrict__ ir,
                                                                        size_t frames, int K, int Lw) {
    const size_t total = frames * Lw;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const size_t f = g / Lw;
        const int tap = (int)(g - f * Lw);
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc = __builtin_fmaf(mags[f * K + k], M[(size_t)k * Lw + tap], acc);
        ir[g] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// time-varying FIR, tiled
// ------------------------------------------------------------------------------------------------
constexpr int FIR_W = 512;          // outputs per wavefront
constexpr int FIR_PASS = 256;       // outputs per pass (64 lanes x 4)
constexpr int FIR_MAX_FRAMES = 80;  // frames staged per workgroup
constexpr int FIR_BW = 4 * FIR_W;    // outputs per workgroup
constexpr int FIR_G_FLOATS = 7168;   // per workgroup: staged frame FIRs
constexpr int FIR_X_FLOATS = 2432;   // per workgroup: staged noise window incl. zero blocks around the signal

__global__ void __launch_bounds__(256) tv_fir_kernel(const float* __restrict__ x,   // [R, N]
                                                   const float* __restrict__ ir,  // [R, T, Lw]
                                                   float* __restrict__ out,       // [R, N]
                                                   int R, int N, int T, int U, int Lw, int delay,
                                                   int windows_per_row, int padl, int nb, int seglen) {
    // One workgroup = FIR_BW consecutive outputs of one row: the frame FIRs and the noise samples that
    // reach them are staged ONCE for the four wavefronts (each then owns FIR_W outputs), which cuts
    // the re-read of impulse responses shared by neighbouring windows from 1.9x to 1.2x.
    __shared__ __attribute__((aligned(16))) float lds[FIR_G_FLOATS + FIR_X_FLOATS + 4 * 256];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int row = blockIdx.x / windows_per_row;
    const int nB0 = (blockIdx.x - row * windows_per_row) * FIR_BW;
    const int n0 = nB0 + wib * FIR_W;             // this wavefront's outputs
    float* G = lds;
    float* Xs = G + FIR_G_FLOATS;
    float* red = Xs + FIR_X_FLOATS + wib * 256;
    const int gstride = nb * 4;                   // floats per staged frame

    // frames whose noise blocks can reach this workgroup's outputs
    const int j_first = max(nB0 + delay - (Lw - 1), 0);
    const int j_last = min(nB0 + FIR_BW - 1 + delay, N - 1);
    const int f_lo = j_first / U;
    const int f_hi = min(j_last / U, T - 1);
    const int nfr = f_hi - f_lo + 1;
    // input blocks (of 4 samples) any lane may touch: zero outside the signal, so the inner loop
    // needs no bounds logic at all
    const int jb_min = (nB0 + delay + 3) / 4 - 4 * seglen;           // may be negative
    const int jb_max = (nB0 + FIR_BW - 64 + delay + 3) / 4 + 15;
    // Staging: every load is issued before the first LDS store (branch-free clamped addresses), so a
    // wavefront pays the HBM/L2 latency once per batch of loads instead of once per load.
    {
        constexpr int XL = (FIR_X_FLOATS + 255) / 256;
        const float* xg = x + (size_t)row * N;
        const int nx = (jb_max - jb_min + 1) * 4;
        const int j0 = 4 * jb_min;
        float xv[XL];
#pragma unroll
        for (int u = 0; u < XL; ++u) xv[u] = xg[min(max(j0 + (int)threadIdx.x + 256 * u, 0), N - 1)];
#pragma unroll
        for (int u = 0; u < XL; ++u) {
            const int q = threadIdx.x + 256 * u, j = j0 + q;
            if (q < nx) Xs[q] = (j >= 0 && j < N) ? xv[u] : 0.0f;
        }
    }
    {
        constexpr int FB = 4;                       // frames per batch (per wavefront)
        constexpr int QL = 5;                       // 64-lane strips per staged frame (gstride <= 320)
        for (int f0 = wib * FB; f0 < nfr; f0 += 4 * FB) {
            float gv[FB][QL];
#pragma unroll
            for (int ff = 0; ff < FB; ++ff) {
                const float* src = ir + ((size_t)row * T + f_lo + min(f0 + ff, nfr - 1)) * Lw;
#pragma unroll
                for (int u = 0; u < QL; ++u) {
                    const int tap = lane + 64 * u - padl;
                    gv[ff][u] = src[min(max(tap, 0), Lw - 1)];
                }
            }
#pragma unroll
            for (int ff = 0; ff < FB; ++ff) {
#pragma unroll
                for (int u = 0; u < QL; ++u) {
                    const int q = lane + 64 * u, tap = q - padl;
                    if (f0 + ff < nfr && q < gstride)
                        G[(f0 + ff) * gstride + q] = (tap >= 0 && tap < Lw) ? gv[ff][u] : 0.0f;
                }
            }
        }
    }
    __syncthreads();

    // ---- compute: 64 outputs per pass.  Lane = (output block a, input segment sg): the ~(Lw + 6) / 4
    // input blocks that reach output block a are split over four lanes, so every FMA carries a tap
    // inside (or one block around) the FIR's support.  The lane -> (sg, a) map follows the way the LDS
    // services a ds_read_b128 (four fixed groups of 16 lanes): each group is one segment, so its 16
    // noise blocks are consecutive (conflict free) and its tap block is one address (broadcast).
    const int bpf = U / 4;                               // input blocks per frame
    const int h = lane >> 5, w = lane & 31;
    const bool g0 = (w < 4) || (w >= 12 && w < 16) || (w >= 20 && w < 28);
    const int sg = 2 * h + (g0 ? 0 : 1);
    const int a = g0 ? (w < 4 ? w : (w < 16 ? w - 8 : w - 12)) : (w < 12 ? w - 4 : (w < 20 ? w - 8 : w - 16));
    for (int ps = 0; ps < FIR_W / 64; ++ps) {
        const int np0 = n0 + ps * 64;
        if (np0 >= N) break;
        const int m0 = np0 + delay;
        const int c0 = (m0 - 3 + padl) / 4;              // exact: (delay - 3 + padl) % 4 == 0, np0 % 4 == 0
        const int jb = (m0 + 3) / 4 + a - sg * seglen;   // first (highest) input block of this lane
        const int bq0 = 4 * (c0 + a - jb);               // float offset of the lower tap block; >= 0, same for all a
        // A lane's seglen (<= U / 4) input blocks lie in at most two frames: the first n1 steps use the
        // frame of jb, the rest the frame before it.  Both tap pointers are per-lane constants, the
        // loop body is branch free (one select per step) and carries no state between steps.
        const int jbc = min(max(jb, 0), (N - 1) / 4);
        const int fA = min(max((4 * jbc) / U, f_lo), f_hi);
        const int n1 = jb - fA * bpf + 1;                // blocks past the end of the signal count as its last frame
        const float* GA = G + (fA - f_lo) * gstride + bq0;
        const float* GB = G + (max(fA - 1, f_lo) - f_lo) * gstride + bq0;
        const float* xp = Xs + 4 * (jb - jb_min);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int i = 0; i < seglen; ++i) {
            const float* gp = (i < n1 ? GA : GB) + 4 * i;
            const float4 lo = *reinterpret_cast<const float4*>(gp);
            const float4 hi = *reinterpret_cast<const float4*>(gp + 4);
            const float4 xv = *reinterpret_cast<const float4*>(xp - 4 * i);
            const float tp[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[e] = __builtin_fmaf(xs[d], tp[e - d + 3], acc[e]);
        }
        // meet the four segments of each output block
        *reinterpret_cast<float4*>(red + (sg * 16 + a) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 16) {
            float4 s0 = *reinterpret_cast<const float4*>(red + lane * 4);
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(red + (k * 16 + lane) * 4);
                s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
            }
            const int n = np0 + 4 * lane;
            if (n < N) *reinterpret_cast<float4*>(out + (size_t)row * N + n) = s0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

__global__ void __launch_bounds__(256) tv_fir_generic_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ ir,
                                                           float* __restrict__ out, int R, int N, int T,
                                                           int U, int Lw, int delay) {
    const size_t total = (size_t)R * N;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int row = (int)(g / N), n = (int)(g - (size_t)row * N);
        const int m = n + delay;
        const int j_lo = max(m - (Lw - 1), 0), j_hi = min(m, N - 1);
        float acc = 0.0f;
        for (int j = j_lo; j <= j_hi; ++j) {
            const int f = min(j / U, T - 1);
            acc = __builtin_fmaf(x[(size_t)row * N + j], ir[((size_t)row * T + f) * Lw + (m - j)], acc);
        }
        out[g] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 uniform noise in [-1, 1)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
}

__global__ void __launch_bounds__(256) uniform_noise_kernel(float* __restrict__ out, size_t n4,
                                                          uint64_t seed, uint64_t offset) {
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n4; g += (size_t)gridDim.x * 256) {
        const uint64_t ctr = offset + g;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
        for (int r = 0; r < 10; ++r) philox_round(c, k);
        float4 v;
        v.x = (float)(c[0] >> 8) * (2.0f / 16777216.0f) - 1.0f;
        v.y = (float)(c[1] >> 8) * (2.0f / 16777216.0f) - 1.0f;
        v.z = (float)(c[2] >> 8) * (2.0f / 16777216.0f) - 1.0f;
        v.w = (float)(c[3] >> 8) * (2.0f / 16777216.0f) - 1.0f;
        reinterpret_cast<float4*>(out)[g] = v;
    }
}

static unsigned stream_grid(size_t total) {
    size_t blocks = (total + 255) / 256;
    const size_t cap = 256 * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

static int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    if (!s || !*s) return dflt;
    return atoi(s);
}

}  // namespace ddspp

using namespace ddspp;

extern "C" {

// ddsp.core.frequency_impulse_response(magnitudes, window_size) as one product with the host-built
// matrix M[K, Lw] (ddsp_piano_amd/core.py: irfft basis x window, shifted to causal form).
// uniq[n_uniq] / mirror[n_uniq] (device int32, may be NULL): the taps to compute and where each
// one is mirrored to (-1: nowhere); NULL means every tap is computed.
int ddspp_fir_from_magnitudes(const float* magnitudes, const float* M, const int* uniq, const int* mirror,
                              int n_uniq, float* ir, size_t frames, int K, int Lw, hipStream_t stream) {
    DDSPP_REQUIRE(magnitudes && M && ir, "fir_from_magnitudes: null buffer");
    DDSPP_REQUIRE(K > 0 && Lw > 0, "fir_from_magnitudes: bad dims");
    DDSPP_REQUIRE(frames < (1ull << 31), "fir_from_magnitudes: too many frames");
    if (frames == 0) return DDSPP_OK;
    const bool tiled = uniq && mirror && n_uniq > 0 && n_uniq <= 256 &&
                       (K == 32 || K == 64 || K == 96 || K == 128) && !env_int("DDSPP_FIR_GENERIC", 0);
    if (tiled) {
        const int fpb = env_int("DDSPP_FIR_FPB", 256);
        const dim3 grid((unsigned)((frames + fpb - 1) / fpb));
        const int nf = (int)frames;
#define DDSPP_FIR_LAUNCH(KK)                                                                              \
        do {                                                                                              \
            if (n_uniq <= 128)                                                                            \
                hipLaunchKernelGGL((fir_from_magnitudes_kernel<KK, 128>), grid, dim3(128), 0, stream,     \
                                   magnitudes, M, uniq, mirror, ir, nf, Lw, n_uniq, fpb);                 \
            else                                                                                          \
                hipLaunchKernelGGL((fir_from_magnitudes_kernel<KK, 256>), grid, dim3(256), 0, stream,     \
                                   magnitudes, M, uniq, mirror, ir, nf, Lw, n_uniq, fpb);                 \
        } while (0)
        if (K == 32) DDSPP_FIR_LAUNCH(32);
        else if (K == 64) DDSPP_FIR_LAUNCH(64);
        else if (K == 96) DDSPP_FIR_LAUNCH(96);
        else DDSPP_FIR_LAUNCH(128);
#undef DDSPP_FIR_LAUNCH
    } else {
        hipLaunchKernelGGL(fir_from_magnitudes_generic_kernel, dim3(stream_grid(frames * Lw)), dim3(256), 0,
                           stream, magnitudes, M, ir, frames, K, Lw);
    }
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// ddsp.core.fft_convolve(audio, impulse_response[B, T, Lw], padding='same', delay_compensation)
// for the framed (time-varying) case: frame_size = hop = U = N / T.
int ddspp_time_varying_fir(const float* audio, const float* impulse_response, float* out, int R, int N,
                           int T, int Lw, int delay_compensation, hipStream_t stream) {
    DDSPP_REQUIRE(audio && impulse_response && out, "time_varying_fir: null buffer");
    DDSPP_REQUIRE(R > 0 && N > 0 && T > 0 && Lw > 0, "time_varying_fir: bad dims");
    DDSPP_REQUIRE(N % T == 0, "time_varying_fir: n_samples=%d must be a multiple of n_frames=%d", N, T);
    const int U = N / T;
    const int delay = delay_compensation < 0 ? (Lw - 1) / 2 - 1 : delay_compensation;
    DDSPP_REQUIRE(delay >= 0, "time_varying_fir: negative delay");
    // tap t of a frame sits at float padl + t of its staged image; padl makes (delay - 3 + padl) a
    // multiple of 4 (aligned 16-byte tap blocks) and leaves >= 2 zero blocks in front, nb leaves zero
    // blocks behind for every block index the inner loop can form (no clamping in the loop)
    const int padl = 8 + ((3 - (delay % 4)) % 4 + 4) % 4;
    const int seglen = (((Lw + 6) / 4 + 1) + 3) / 4;
    const int nb_need = (padl + Lw + 3) / 4 + 1;
    const int nb = (4 * seglen + 4 > nb_need ? 4 * seglen + 4 : nb_need) + 1;
    const int frames_max = (FIR_BW + Lw + U - 2) / U + 2;
    const bool tiled = (U % 4 == 0) && (N % 4 == 0) && ((uintptr_t)audio % 16 == 0) &&
                       ((uintptr_t)out % 16 == 0) && frames_max <= FIR_MAX_FRAMES &&
                       frames_max * nb * 4 <= FIR_G_FLOATS && (FIR_BW / 4 + 16 + 4 * seglen + 2) * 4 <= FIR_X_FLOATS &&
                       nb * 4 <= 320 && seglen <= U / 4 &&
                       !env_int("DDSPP_FIR_GENERIC", 0);
    if (tiled) {
        const int wpr = (N + FIR_BW - 1) / FIR_BW;
        const long long tasks = (long long)R * wpr;
        DDSPP_REQUIRE(tasks < (1ll << 31), "time_varying_fir: too many tasks");
        hipLaunchKernelGGL(tv_fir_kernel, dim3((unsigned)tasks), dim3(256), 0, stream, audio,
                           impulse_response, out, R, N, T, U, Lw, delay, wpr, padl, nb, seglen);
    } else {
        hipLaunchKernelGGL(tv_fir_generic_kernel, dim3(stream_grid((size_t)R * N)), dim3(256), 0, stream,
                           audio, impulse_response, out, R, N, T, U, Lw, delay);
    }
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// Same operator through the even/odd tables (host: ddsp_piano_amd/core.py::_fir_eo_tables);
// K must be even and a multiple of 4, NJ <= 64.
int ddspp_fir_from_magnitudes_eo(const float* magnitudes, const float* CE, const float* CO, const int* tap_idx,
                                 const float* tap_we, const float* tap_wo, float* ir, size_t frames, int K,
                                 int Lw, int NJ, hipStream_t stream) {
    DDSPP_REQUIRE(magnitudes && CE && CO && tap_idx && tap_we && tap_wo && ir, "fir_from_magnitudes_eo: null buffer");
    DDSPP_REQUIRE(K > 0 && K % 4 == 0 && NJ > 0 && NJ <= 64 && Lw > 0, "fir_from_magnitudes_eo: bad dims");
    DDSPP_REQUIRE(K == 32 || K == 64 || K == 96 || K == 128, "fir_from_magnitudes_eo: unsupported band count %d", K);
    DDSPP_REQUIRE(frames < (1ull << 31), "fir_from_magnitudes_eo: too many frames");
    DDSPP_REQUIRE((uintptr_t)magnitudes % 16 == 0, "fir_from_magnitudes_eo: magnitudes must be 16-byte aligned");
    if (frames == 0) return DDSPP_OK;
    const int fpb = 64;
    const dim3 grid((unsigned)((frames + fpb - 1) / fpb)), block(256);
    const size_t lds = (size_t)fpb * K * sizeof(float);
    const int nf = (int)frames;
    if (K == 32) hipLaunchKernelGGL(fir_eo_kernel<16>, grid, block, lds, stream, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, ir, nf, Lw, NJ, fpb);
    else if (K == 64) hipLaunchKernelGGL(fir_eo_kernel<32>, grid, block, lds, stream, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, ir, nf, Lw, NJ, fpb);
    else if (K == 96) hipLaunchKernelGGL(fir_eo_kernel<48>, grid, block, lds, stream, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, ir, nf, Lw, NJ, fpb);
    else hipLaunchKernelGGL(fir_eo_kernel<64>, grid, block, lds, stream, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, ir, nf, Lw, NJ, fpb);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// U(-1, 1) noise, Philox4x32-10(counter = offset + i / 4, key = seed); n % 4 == 0.
int ddspp_uniform_noise(float* out, size_t n, uint64_t seed, uint64_t offset, hipStream_t stream) {
    DDSPP_REQUIRE(out, "uniform_noise: null buffer");
    DDSPP_REQUIRE(n % 4 == 0, "uniform_noise: n must be a multiple of 4");
    if (n == 0) return DDSPP_OK;
    hipLaunchKernelGGL(uniform_noise_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, stream, out, n / 4, seed,
                       offset);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

}  // extern "C"
