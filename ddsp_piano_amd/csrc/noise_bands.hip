// NoiseBandNetSynth.get_signal (ddsp_piano/modules/filtered_noise_synth.py:213-262): a bank of fixed, loopable
// band-limited noises, each modulated by a frame-rate amplitude that is linearly upsampled chunk by chunk, summed
// over the bands:
//   audio[b, n] = sum_k noise_bands[(n mod noise_len - shift) mod noise_len, k] *
//                       (a[b, lo[n], k] + (a[b, hi[n], k] - a[b, lo[n], k]) * w[n])
// lo / hi / w are ddsp.core.resample(method='linear') of each noise_len-sample chunk of frames (the reference
// resamples chunk by chunk, :238-259, so the interpolation restarts -- last frame held -- at every chunk boundary, and
// stretches a short last chunk to a full one); the host builds them with the library's legacy-bilinear tables.
// One wavefront per (row, 32 samples): lane = band (coalesced 256-byte reads of a noise-band row and of the two
// amplitude frames), per sample each lane multiplies its bands, the products of the 64 lanes are summed through the
// [32][64] LDS tile the oscillator bank uses (conflict-free transposed reads).
#include "ddspp_common.h"

namespace ddspp {

constexpr int NB_TILE = 32, NB_STRIDE = 68;

__global__ void __launch_bounds__(256) noise_bands_kernel(const float* __restrict__ amplitudes,   // [R, T, K]
                                                        const float* __restrict__ noise_bands,  // [noise_len, K]
                                                        const int* __restrict__ lo, const int* __restrict__ hi,
                                                        const float* __restrict__ w,            // [N]
                                                        float* __restrict__ out,                // [R, N]
                                                        int R, int T, int K, int N, int noise_len, int shift) {
    __shared__ float tiles[4][NB_TILE * NB_STRIDE];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    float* tile = tiles[wib];
    const int tiles_per_row = (N + NB_TILE - 1) / NB_TILE;
    const size_t task = (size_t)blockIdx.x * 4 + wib;
    if (task >= (size_t)R * tiles_per_row) return;
    const int row = (int)(task / tiles_per_row), n0 = (int)(task - (size_t)row * tiles_per_row) * NB_TILE;
    const float* arow = amplitudes + (size_t)row * T * K;
    for (int s = 0; s < NB_TILE; ++s) {
        const int n = min(n0 + s, N - 1);
        int pos = (n % noise_len) - shift;                   // tf.roll(noise_bands, shift, axis=1)
        pos %= noise_len;
        if (pos < 0) pos += noise_len;
        const float wn = w[n];
        const float* al = arow + (size_t)lo[n] * K;
        const float* ah = arow + (size_t)hi[n] * K;
        const float* nb = noise_bands + (size_t)pos * K;
        float acc = 0.0f;
        for (int k = lane; k < K; k += 64) {
            const float a0 = al[k], a1 = ah[k];
            const float a = a0 + (a1 - a0) * wn;             // legacy bilinear (core.resample)
            acc += nb[k] * a;
        }
        tile[s * NB_STRIDE + lane] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int col = lane & 31, half = lane >> 5;
    const float* src = tile + col * NB_STRIDE + half * 32;
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) sum += src[i];
    sum += __shfl_xor(sum, 32);
    if (lane < 32 && n0 + lane < N) out[(size_t)row * N + n0 + lane] = sum;
}

}  // namespace ddspp

using namespace ddspp;

extern "C" {

// NoiseBandNetSynth.get_signal -- filtered_noise_synth.py:213-262.  amplitudes[R,T,K] (after scale_fn),
// noise_bands[noise_len,K], lo/hi/w[N]: the chunk-wise linear resampling tables, shift: the roll of the noise bands.
int ddspp_noise_bands(const float* amplitudes, const float* noise_bands, const int* lo, const int* hi, const float* w,
                      float* audio, int R, int T, int K, int N, int noise_len, int shift, hipStream_t stream) {
    DDSPP_REQUIRE(amplitudes && noise_bands && lo && hi && w && audio, "noise_bands: null buffer");
    DDSPP_REQUIRE(R > 0 && T > 0 && K > 0 && N > 0 && noise_len > 0, "noise_bands: bad dims");
    const size_t tasks = (size_t)R * ((N + NB_TILE - 1) / NB_TILE);
    DDSPP_REQUIRE((tasks + 3) / 4 < (1ull << 31), "noise_bands: too many samples");
    hipLaunchKernelGGL(noise_bands_kernel, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, stream, amplitudes, noise_bands,
                       lo, hi, w, audio, R, T, K, N, noise_len, shift);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

}  // extern "C"
