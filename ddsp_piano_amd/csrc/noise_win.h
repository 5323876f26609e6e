// Windowed ("one lane = one frame") time-varying FIR kernels, csrc/noise_win.hip; called from the entry points in noise.hip.
#pragma once
#include "ddspp_common.h"

namespace ddspp {

// Geometry of a window (see noise_win.hip): everything the kernels need beyond their template parameters.
struct WinGeom {
    int U;        // hop = N / T
    int delay;    // crop_and_compensate_delay's start
    int padl;     // float offset of tap 0 inside a frame image
    int gs;       // floats between two frame images (gs / 4 odd)
    int gshift;   // floats of the first image's lower gap that are not allocated (multiple of 4)
    int nsteps;   // input blocks a lane walks
    int q_hi0;    // block index (relative to the lane's frame) of phase 0's first step
    int RL, RH;   // frames before the first / after the last computed one that a window needs
    int W;        // frames a window computes
    int wpr;      // windows per row
    int opl;      // outputs per lane
    int bpf;      // U / 4
    int trim;     // the walk's first and last steps are the exact triangles fir_win_step can leave out
    int Lw;       // taps of an impulse response
    // Round 6: the noise drawn inside the kernel instead of read from a [R, N] tensor (ddspp_frequency_filter_eo_voices_drawn):
    // block jb of row r is philox_uniform4(draw_seed, draw_offset + r N / 4 + jb) -- what ddspp_uniform_noise(seed, offset)
    // writes at that place of a [R, N] tensor, bit for bit; the tensor (0.29 GB written and read back per step at config 3)
    // never exists.  Set by launch_win_fused, not by win_geometry.
    int draw_on;
    unsigned long long draw_seed, draw_offset;
};

// Philox4x32-10, counter (ctr, 0), key = seed -> four U(-1, 1) floats (24 random bits each): the library's stand-in for the
// reference's unseeded tf.random.uniform (filtered_noise_synth.py:39-40).  One definition for ddspp_uniform_noise and for
// the kernels that draw their own noise.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    // the 64-bit products written as such: one v_mad_u64_u32 each instead of a v_mul_hi_u32 + v_mul_lo_u32 pair -- the same
    // numbers, 102 against 129 ns per counter and SIMD (tools/ubench/philox_rates, profiles/r06_ubench.txt)
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)c[0], p1 = (uint64_t)M1 * (uint64_t)c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
}
__device__ __forceinline__ float4 philox_uniform4(uint64_t seed, uint64_t ctr) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
    float4 v;
    v.x = (float)(c[0] >> 8) * (2.0f / 16777216.0f) - 1.0f;
    v.y = (float)(c[1] >> 8) * (2.0f / 16777216.0f) - 1.0f;
    v.z = (float)(c[2] >> 8) * (2.0f / 16777216.0f) - 1.0f;
    v.w = (float)(c[3] >> 8) * (2.0f / 16777216.0f) - 1.0f;
    return v;
}

bool win_fused_supported(int N, int T, int K, int Lw, int delay, WinGeom* g);
bool win_tvfir_supported(int N, int T, int Lw, int delay, WinGeom* g);
int launch_win_fused(const float* audio, const float* magnitudes, const float* CE, const float* CO, const int* tap_idx,
                     const float* tap_we, const float* tap_wo, float* out, float* out_last, int R, int N, int T, int K,
                     int NJ, const WinGeom& g, float bias, const ScaleFn& sf, int vq, int n_voices, int voice_major,
                     hipStream_t stream, bool draw = false, unsigned long long draw_seed = 0, unsigned long long draw_offset = 0);
int launch_win_tvfir(const float* audio, const float* ir, float* out, int R, int N, int T, int Lw, const WinGeom& g,
                     hipStream_t stream);

}  // namespace ddspp
