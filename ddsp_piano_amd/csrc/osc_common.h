// Shared by the oscillator-bank translation units (oscillator.hip, bank_compact.hip).
#pragma once
#include "ddspp_common.h"

namespace ddspp {

struct OscParams {
    // materialised source: cos_oscillator_bank(frequency_envelopes, amplitude_envelopes)
    const float* __restrict__ fe;      // [R, N, H]
    const float* __restrict__ ae;      // [R, N, H]
    // fused source: frame-rate controls of harmonic_synthesis / MultiInharmonic.get_signal
    const float* __restrict__ f0;      // [R, T, S]
    const float* __restrict__ amp;     // [R, T]
    const float* __restrict__ hd;      // [R, T, H]
    const float* __restrict__ shifts;  // [R, T, H] (may be null: no shifts, or shifts from `inh`)
    const float* __restrict__ inh;     // [R, T] raw inharm_coef: with shifts == null the kernels form harmonic_shifts themselves
    const int* __restrict__ audible;   // [R, T] leading non-silent harmonics per frame (pre-pass: may be null)
    const int* __restrict__ rowmax;    // [R] max over the frames of a row of `audible` (low 16 bits), or null: the kernels reduce it themselves
    int dbg_noflags;                   // DDSPP_OSC_NO_FLAGS=1: ignore bit 16 of audible, stream the controls instead (A/B switch)
    const float* __restrict__ wlin;    // [N]   legacy-bilinear interpolation weight per sample
    const float* __restrict__ whann;   // [2U]  tf.signal.hann_window(2U)
    float* __restrict__ out;           // [R, N] (sum) or [R, N, V]
    float* __restrict__ ework;         // [R, npre, VP]  chunk end phase mod 2pi
    float* __restrict__ echunk;        // sectioned memo pre-pass: [R, npre, VP] chunk end phases (ework then is astart)
    int nsec;                          // ... and its sections per (row, group)
    const float* __restrict__ astart;  // [R, spans, VP] running offset sum at span start
    float* __restrict__ partial;       // [R, groups, N] per-group audio when groups > 1
    int R, N, T, U, H, S, V, VP;
    int groups, vgrp;                  // oscillators of a row are split over `groups` wavefronts
    int spans, cps, nchunks, npre;
    float sr, rsr, nyq;
    int fastdiv;                       // sample rate is in the exhaustively checked list
    // ddsp.core.angular_cumsum adds the running sum A of the chunks' end phases to every chunk.  Recalled detail
    // (`angular_offsets`): 0 = `offsets = tf.cumsum(offsets, axis=1) % (2 pi)` (wrapped, the default),
    // 1 = the sum is added as it is (DDSPP_ANGULAR_OFFSETS_PLAIN=1 / ddspp_set_option).  A itself is the same float32
    // running sum either way.
    int off_plain;
    // compacted polyphonic mode (osc_kernel<1, true, MODE_MAIN, true, true>): R = segments, each with
    // P voices; only oscillators with a non-zero amplitude somewhere in the span are given a lane
    const int* __restrict__ nk;        // [B, spans, P] audible harmonics per voice and span
    int* __restrict__ wcount;          // [B, spans]    wavefronts that actually produced a partial row
    int P, wmax, nslots;               // voices per segment, partial rows per segment, workgroups per (segment, span)
    int vmajor;                        // compact mode: rows are [P, B] (voice major) instead of [B, P]
    // compacted bank (bank_compact.hip): slots [0, wmax_a) carry the audible oscillators of voices [0, P - split_last),
    // slots [wmax_a, wmax) those of the last voice (split_last = 1: the caller wants that voice's stem on its own)
    int split_last, wmax_a;
    int pair;                          // S = 2 and two oscillators per lane: a lane carries both sub-strings of a (voice, harmonic) (bank_slot<2, false, PAIR>; DDSPP_OSC_PAIR)
    int half_slots;                    // a 128-oscillator slot with <= 64 oscillators left runs the 64-oscillator body (DDSPP_OSC_HALF_SLOTS)
    int held_skip;                     // frame boundaries of held notes skip the frequency / classification work (DDSPP_OSC_HELD_SKIP)
    float* __restrict__ out_last;      // [B, N] the last voice's stem (split_last = 1), `out` then holds the other voices' sum
    int stem_blocks;                   // every voice's stem (bank_stems_kernel): `out` is [B, stem_blocks, N], a row per 32-entry block of the packed list
    // streaming: the float32 running sum of chunk end phases (ddsp.core.angular_cumsum's cumsum over chunks) each
    // oscillator starts from, [rows, V]; null = 0 (a signal that starts here)
    const float* __restrict__ state_in;
    // the span-start walk normally leaves out 64-oscillator groups that are silent in every frame of the call (nothing
    // reads their start phases); need_all = 1 walks them too: a carried phase state must cover every oscillator
    int need_all;
    // Compacted scan of moving chunks (round 5).  The span starts need the end phase of EVERY chunk before them.  Chunks of
    // held notes are memoised by the pre-pass (osc_prepass_fused_kernel); chunks in which some frequency moves have to be
    // scanned sample by sample, and the pre-pass does that with one wavefront per (row, 64 harmonics) -- 1.7 x the lanes a
    // piano's audible partials fill.  With skip_moving the pre-pass leaves those chunks alone and bank_scan_kernel scans
    // them with the bank's packing (lanes only for harmonics below the row's audible maximum, voices back to back):
    // scan_tasks [R rows, npre]: 1 where the row's frequencies move in the chunk, scan_ntasks[0] == scan_call when any row moves
    // anywhere (both written by osc_count_rows_kernel, nothing to zero); scan_slots: slots of 64 VPL oscillators a segment may need.
    int* __restrict__ scan_tasks;
    int* __restrict__ scan_ntasks;
    int scan_slots, skip_moving, scan_call;
    // SurrogateAdditive (surrogate_synth.py:76-95): per-harmonic exponential decay of the amplitude envelopes,
    // |decays[t, k]| ** (decay_time[t] * U + n % U) with t = n / U (the frame's values repeated, not interpolated); null = none
    const float* __restrict__ decays;      // [R, T, H]
    const float* __restrict__ decay_time;  // [R, T]
};

enum { MODE_MAIN = 0, MODE_PREPASS = 1, MODE_PLAIN = 2 };

constexpr int BLK = 8;        // samples per unrolled block; divides U and the 1000-sample chunk
// `wlin` as the frame-walking kernels take it (ddspp_walk_weights_host): the fractional part of float32(n) * float32(T / N),
// or this mark where that product rounds up to the next whole frame -- far into a long file (frame 131 073 at hop 96) the
// reference's resize takes row t + 1 with weight 0 for the last sample(s) of frame t, i.e. x[t + 1] itself, which
// x0 + (x1 - x0) w cannot produce exactly for any w.  A fractional part is never 1, marked samples are a suffix of their
// frame's last block: one scalar compare per frame pair finds them.
constexpr float WALK_NEXT_ROW = 1.0f;
constexpr int TILE = 32;      // samples per LDS reduction tile
constexpr int TSTRIDE = 68;   // words per tile row: 16-byte aligned rows, 17 quads apart -> ds_read_b128 conflict free

// harmonic_shifts[t, k] of get_inharmonic_freq (inharm_synth.py:37-44) for harmonic number m = k + 1, from the raw
// inharm_coef (clamped as InHarmonic.get_controls does, :183).  Every op separately rounded, the square root correctly
// rounded: bit for bit what ddspp_inharmonic_controls writes into harmonic_shifts_out, so a kernel may form the shifts
// itself (per lane and frame) instead of reading a [R, T, H] tensor back.
__device__ __forceinline__ float shift_from_inharm(float inharm_raw, float m) {
    const float inharm = fmaxf(inharm_raw, 0.0f);
    float g = m * m;                               // tf.math.pow(int_multiplier, 2)        :37
    g = g * inharm + 1.0f;                         //                                        :38
    g = sqrtf(g);                                  //                                        :39
    return g - 1.0f;                               //                                        :44
}

// the chunk offset from the running sum of chunk end phases (see OscParams::off_plain)
__device__ __forceinline__ float chunk_offset(float asum, int off_plain) {
    const float w = mod_2pi(asum);
    return off_plain ? asum : w;
}

template <bool FAST>
__device__ __forceinline__ float omega_of(float fe, float sr, float rsr) {
    float om = fe * DDSPP_TWO_PI_F32;             // inharm_synth.py:69
    if (FAST) return div_const(om, sr, rsr);      // inharm_synth.py:70, exact (see ddspp_common.h)
    return om / sr;
}


constexpr int PRE_W = 256;              // floats of LDS per wavefront: the interpolation weights of the next PRE_W samples (scans)

// BLK samples of the phase scan `ph += omega(x0 + (x1 - x0) * w[i])` for one oscillator per lane, in stages that keep
// BLK independent chains in flight: interpolate all BLK frequencies, scale them all, divide them all, and only then
// the BLK dependent adds.  Written sample after sample the compiler funnels every omega through the same two
// temporaries -- one dependent chain of forty instructions per step, and a dependent wave64 instruction issues every
// 4.4 cycles instead of 2.3 whatever the other wavefronts of the SIMD do (tools/ubench/valu_latency): the
// moving-frequency pre-pass ran at a third of the issue rate.  srv / rsrv: sample rate and its reciprocal in VECTOR
// registers (an SGPR operand in a VOP3 fma costs 4.2 cycles).  Arithmetic and order are those of omega_of.
template <bool FAST>
__device__ __forceinline__ float scan_block_staged(float ph, float x0, float x1, const float* w, float srv, float rsrv,
                                                   bool next_row = false) {
    float om[BLK], q[BLK];
    const float dx = x1 - x0;
#pragma unroll
    for (int i = 0; i < BLK; ++i) om[i] = dx * w[i];
#pragma unroll
    for (int i = 0; i < BLK; ++i) om[i] = x0 + om[i];
    if (next_row) {                       // wave-uniform: the frame's last block holds marked samples (osc_common.h)
#pragma unroll
        for (int i = 0; i < BLK; ++i) om[i] = (w[i] == WALK_NEXT_ROW) ? x1 : om[i];
    }
#pragma unroll
    for (int i = 0; i < BLK; ++i) om[i] = om[i] * DDSPP_TWO_PI_F32;          // inharm_synth.py:69
    if (FAST) {                                                              // :70, div_const (ddspp_common.h)
#pragma unroll
        for (int i = 0; i < BLK; ++i) q[i] = om[i] * rsrv;
#pragma unroll
        for (int i = 0; i < BLK; ++i) om[i] = __builtin_fmaf(-q[i], srv, om[i]);
#pragma unroll
        for (int i = 0; i < BLK; ++i) om[i] = __builtin_fmaf(om[i], rsrv, q[i]);
    } else {
#pragma unroll
        for (int i = 0; i < BLK; ++i) om[i] = om[i] / srv;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < BLK; ++i) ph = ph + om[i];
    __builtin_amdgcn_sched_barrier(0);
    return ph;
}

__device__ __forceinline__ float in_vgpr(float x) {      // a wave-uniform value copied into a vector register
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
    return v;
}

// compacted polyphonic bank: launches of bank_compact.hip (p.out = partial rows, see ddspp_polyphonic_additive)
void launch_bank_compact(const OscParams& p, int vpl, hipStream_t stream);
void launch_bank_slot_sum(const OscParams& p, float* audio, float* audio_last, hipStream_t stream);
// every voice's stem: the bank with the harmonic sum stopped at voice boundaries + the sum of a voice's blocks -> stems [B * P, N]
void launch_bank_stems(const OscParams& p, int vpl, float* stems, hipStream_t stream);
// compacted scan of the chunks whose frequencies move (bank_scan_kernel; p as for launch_bank_compact + the scan_* fields)
void launch_bank_scan(const OscParams& p, int vpl, hipStream_t stream);
// the HBM-roofline form of cos_oscillator_bank (osc_stream.hip): materialised envelopes, angular cumsum, summed, H % 64 == 0
bool osc_stream_applies(const OscParams& p);
void launch_osc_stream(const OscParams& p, hipStream_t stream);

}  // namespace ddspp
