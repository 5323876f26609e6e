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
};

bool win_fused_supported(int N, int T, int K, int Lw, int delay, WinGeom* g);
bool win_tvfir_supported(int N, int T, int Lw, int delay, WinGeom* g);
int launch_win_fused(const float* audio, const float* magnitudes, const float* CE, const float* CO, const int* tap_idx,
                     const float* tap_we, const float* tap_wo, float* out, float* out_last, int R, int N, int T, int K,
                     int NJ, const WinGeom& g, float bias, const ScaleFn& sf, int vq, int n_voices, int voice_major,
                     hipStream_t stream);
int launch_win_tvfir(const float* audio, const float* ir, float* out, int R, int N, int T, int Lw, const WinGeom& g,
                     hipStream_t stream);

}  // namespace ddspp
