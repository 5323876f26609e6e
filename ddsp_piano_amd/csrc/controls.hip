// Frame-rate control conditioning (get_controls of the processors) and the signal mixer.
//
// Replaces InHarmonic.get_controls / MultiInharmonic.get_controls
// (ddsp_piano/modules/inharm_synth.py:167-219, :254-270), get_inharmonic_freq (:20-46),
// ddsp.synths.FilteredNoise.get_controls, the scale functions ddsp.core.exp_sigmoid and exp_tanh
// (inharm_synth.py:13-17) and MultiAdd.get_signal (:308-309).
// One DPP row of 16 lanes conditions one (row, frame): its lanes run over the harmonics in steps of 16 (64-byte pieces
// of harmonic_distribution[row, t, :]), the normalisation sum is a reduction over the row.
#include "ddspp_common.h"

namespace ddspp {

// DPP data-path moves within a row of 16 lanes (quad swaps, half-row and row mirrors): four of them leave every lane of
// a row with the row's sum.  (ds_bpermute round trips -- what __shfl_xor compiles to -- were a dependent chain of LDS
// latencies in every frame of the get_controls kernel.)
template <int CTRL>
__device__ __forceinline__ float dpp_take(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

struct InharmParams {
    const float* __restrict__ amplitudes;             // [R, T]      raw
    const float* __restrict__ harmonic_distribution;  // [R, T, H]   raw
    const float* __restrict__ inharm_coef;            // [R, T]
    const float* __restrict__ f0_hz;                  // [R, T, S]
    float* __restrict__ amp_out;                      // [R, T]
    float* __restrict__ hd_out;                       // [R, T, H]
    float* __restrict__ shifts_out;                   // [R, T, H]
    int* __restrict__ count_out;                      // [R, T] audible leading harmonics per frame (+ bit 16: frequencies moved), or null
    float* __restrict__ shifts_last;                  // [R / P, T, H] harmonic_shifts of every segment's LAST voice only, or null
    int P, vmajor;                                    // rows are [B, P] (or [P, B] with vmajor) when shifts_last is given
    int R, T, H, S;
    float nyquist, min_frequency, n_substrings;
    int normalize_after_nyquist_cut, normalize_below_nyquist;
    ScaleFn scale;
    // Round 6 (lean kernel only): hd_out of a frame is written only below the frame's audible count, in whole groups of 16
    // (k < 16 ceil(count / 16)); the rest of the row keeps whatever the buffer held.  Every harmonic at or above the count has
    // amp_out * hd_out == 0 by the count's definition, and the compacted bank -- the only reader on that route -- takes them as 0
    // from the count itself (bank_compact.hip, frame_request).  Rows of a segment's last voice (shifts_last given) are
    // written whole: the outputs dictionary keeps that voice's controls.  At a piano's note mix two thirds of the [R, T, H]
    // tensor are never written.
    int hd_sparse;
};

// A frame is conditioned by ONE DPP ROW (16 lanes), lane i of the row taking harmonics i, i + 16, i + 32, ...; a
// wavefront takes CTL_PASS x 4 consecutive frames of the row-major [R * T] frame list, four at a time (CTL_PASS = 2 up
// to 128 harmonics, 1 above: registers).  Against round 2's
// "64 lanes run over the harmonics of one frame": (i) the per-frame scalar work (the amplitude's scale function, the
// row sum, the audible count) is issued once per four frames; (ii) get_controls cuts every harmonic at or above Nyquist
// (:200-208), and a group of 16 harmonics whose FIRST is already there is cut whole whatever its raw values (the
// frequencies grow with the harmonic number, the inharmonicity factor is >= 1): with the cut before the normalisation
// (the default flags) such a group contributes nothing to the sum either, so its raw values are never READ and its
// scale function, square root and division are skipped when the four frames agree (they are neighbours on a note).
// A piano has about a third of its partials below Nyquist; groups of 64 could only ever skip harmonics 65..128.
// The division by the frame's sum is one IEEE reciprocal per frame and Markstein's correction step per element (q =
// x * r, q + fma(-q, d, x) * r: the correctly rounded quotient but for rare last-bit cases; the sum itself is added up in
// another order than numpy's or TensorFlow's).
// All loads of the wavefront are issued before the first use: the scalars of its 8 frames, then (they decide what is
// read) the raw distributions.

__device__ __forceinline__ float row_sum(float v) {          // over the 16 lanes of a DPP row, every lane gets it
    v += dpp_take<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_take<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_take<0x141>(v);       // row_half_mirror
    v += dpp_take<0x140>(v);       // row_mirror
    return v;
}
template <int CTRL>
__device__ __forceinline__ int dpp_take_int(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ int row_max(int v) {
    v = max(v, dpp_take_int<0xB1>(v));
    v = max(v, dpp_take_int<0x4E>(v));
    v = max(v, dpp_take_int<0x141>(v));
    v = max(v, dpp_take_int<0x140>(v));
    return v;
}

template <int NJ, int CTL_PASS>
__device__ __forceinline__ void inharmonic_controls_body(const InharmParams& p) {
    const int lane = threadIdx.x & 63, sub = lane & 15, rowi = lane >> 4;
    const size_t nframes = (size_t)p.R * p.T;
    const size_t wave0 = ((size_t)blockIdx.x * 4 + (size_t)wave_uniform(threadIdx.x >> 6)) * (4 * CTL_PASS);
    if (wave0 >= nframes) return;
    const int H = p.H;
    const bool cut_first = p.normalize_below_nyquist && p.normalize_after_nyquist_cut;
    float raw_f0[CTL_PASS], raw_in[CTL_PASS], raw_amp[CTL_PASS];
#pragma unroll
    for (int u = 0; u < CTL_PASS; ++u) {
        const size_t fr = min(wave0 + 4 * u + rowi, nframes - 1);
        raw_f0[u] = p.f0_hz[fr * p.S];                                  // f0_hz[..., 0:1]  (:264)
        raw_in[u] = p.inharm_coef[fr];
        raw_amp[u] = p.amplitudes[fr];
    }
    // (row, frame in row) of this lane's frames -- only the last voice's shifts are kept when shifts_out is null.  32-bit
    // arithmetic (R * T < 2^31 checked by the host), one division per wavefront.
    unsigned row0 = 0, tt0 = 0, last_lo = 0;
    if (p.shifts_last) {
        row0 = (unsigned)wave0 / (unsigned)p.T;
        tt0 = (unsigned)wave0 - row0 * (unsigned)p.T;
        last_lo = (unsigned)(p.R / p.P) * (unsigned)(p.P - 1);          // voice major: first row of the last voice
    }
    unsigned dead[CTL_PASS];           // bit j: harmonics 16 j .. 16 j + 15 of the lane's frame are cut whole
    float raw_hd[CTL_PASS][NJ];
#pragma unroll
    for (int u = 0; u < CTL_PASS; ++u) {
        const size_t frame = wave0 + 4 * u + rowi;
        unsigned d = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (cut_first && raw_f0[u] * (float)(16 * j + 1) >= p.nyquist) d |= 1u << j;
        if (frame >= nframes) d = ~0u;
        dead[u] = d;
        const float* src = p.harmonic_distribution + min(frame, nframes - 1) * H;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = sub + 16 * j;
            raw_hd[u][j] = (k < H && !((d >> j) & 1)) ? src[k] : 0.0f;
        }
    }
#pragma unroll
    for (int u = 0; u < CTL_PASS; ++u) {
        const size_t frame = wave0 + 4 * u + rowi;
        const bool live = frame < nframes;
        const float f0 = raw_f0[u];
        const float inharm = fmaxf(raw_in[u], 0.0f);                    // :183
        float amp = apply_scale(p.scale, raw_amp[u]);                   // :185
        bool is_last = false;
        unsigned row = row0, tt = tt0 + 4 * (unsigned)u + (unsigned)rowi;
        if (p.shifts_last) {
            while (tt >= (unsigned)p.T) {
                tt -= (unsigned)p.T;
                ++row;
            }
            is_last = live && (p.vmajor ? row >= last_lo : (row % (unsigned)p.P) == (unsigned)p.P - 1);
        }
        const bool want_shift = live && (p.shifts_out != nullptr || is_last);
        float hd[NJ], shift[NJ];
        unsigned above = 0;                                                   // bit j: freq[j] >= nyquist
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = sub + 16 * j;
            const bool dj = (dead[u] >> j) & 1;
            hd[j] = 0.0f;
            shift[j] = 0.0f;
            if (__all(dj && !want_shift)) continue;                           // nothing to do for the four frames
            if (NJ > 8) __builtin_amdgcn_sched_barrier(0);                    // (wide rows: one group after the other, see below)
            if (k < H && (!dj || want_shift)) {
                const float m = (float)(k + 1);
                float g = m * m;                       // tf.math.pow(int_multiplier, 2)        :37
                g = g * inharm + 1.0f;                 //                                        :38
                g = sqrtf(g);                          //                                        :39
                shift[j] = g - 1.0f;                   //                                        :44  (= osc_common.h shift_from_inharm)
                if (!dj) {
                    hd[j] = apply_scale(p.scale, raw_hd[u][j]);                             // :186
                    const float freq = (f0 * m) * g;   // f0_hz * int_multiplier * inharm_factor :42
                    if (freq >= p.nyquist) above |= 1u << j;
                    sum += hd[j];
                }
            }
        }
        if (p.normalize_after_nyquist_cut == 0) {                            // :194-198   (2 = never: SurrogateAdditive)
            const float tot = row_sum(sum);
            const float den = tot == 0.0f ? 1e-7f : tot;                     // core.safe_divide
            const float rden = 1.0f / den;
            const bool tame = __all(den > 1e-30f && den < 1e30f);           // (else: the IEEE division, whatever it gives)
            sum = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                hd[j] = tame ? div_const(hd[j], den, rden) : hd[j] / den;
                sum += hd[j];
            }
        }
        if (p.normalize_below_nyquist) {                                     // :200-208
            sum = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if ((above >> j) & 1) hd[j] = 0.0f;                          // core.remove_above_nyquist
                sum += hd[j];
            }
            amp = amp * (f0 > p.min_frequency ? 1.0f : 0.0f);
        }
        if (p.normalize_after_nyquist_cut == 1) {                            // :210-214
            const float tot = row_sum(sum);
            const float den = tot == 0.0f ? 1e-7f : tot;
            const float rden = 1.0f / den;
            const bool tame = __all(den > 1e-30f && den < 1e30f);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (__all((dead[u] >> j) & 1)) continue;
                if (!((dead[u] >> j) & 1)) hd[j] = tame ? div_const(hd[j], den, rden) : hd[j] / den;   // (a cut group stays 0)
            }
        }
        amp = amp / p.n_substrings;                                          // :269 (1.0 for InHarmonic)
        if (live) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int k = sub + 16 * j;
                if (k < H) {
                    p.hd_out[frame * H + k] = hd[j];
                    if (p.shifts_out) p.shifts_out[frame * H + k] = shift[j];    // null: the oscillator kernels form them from inharm_coef
                }
            }
            if (is_last) {             // what the outputs dictionary of the reference's DAG keeps: the last voice's controls
                const unsigned b = p.vmajor ? row - last_lo : row / (unsigned)p.P;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int k = sub + 16 * j;
                    if (k < H) p.shifts_last[((size_t)b * p.T + tt) * H + k] = shift[j];
                }
            }
            if (sub == 0) p.amp_out[frame] = amp;
        }
        if (p.count_out) {
            // 1 + index of the last harmonic whose sample-rate amplitude amp * hd is not zero in this frame
            // (what ddspp_polyphonic_additive needs to know to skip the silent top of the harmonic range)
            int last = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int k = sub + 16 * j;
                if (k < H && amp * hd[j] != 0.0f) last = k + 1;
            }
            last = row_max(last);
            if (live && sub == 0) p.count_out[frame] = last;        // bit 16 is added by frames_moved_kernel
        }
    }
}

// More than 256 harmonics (no shipped configuration; up to 512): the register-resident body above would keep three arrays
// of 32 values per lane and spilled 7 400 registers' worth.  This one keeps nothing: the harmonics
// of a frame are walked twice, 16 at a time -- once for the sum the normalisation divides by, once more to form, cut,
// normalise and store them -- at the price of evaluating the scale function twice.  Same arithmetic per element; the sum
// is added up in the same lane order (j ascending, then the DPP row sum).
__device__ __forceinline__ void inharmonic_controls_wide_body(const InharmParams& p) {
    const int lane = threadIdx.x & 63, sub = lane & 15, rowi = lane >> 4;
    const size_t nframes = (size_t)p.R * p.T;
    const size_t frame = ((size_t)blockIdx.x * 4 + (size_t)(threadIdx.x >> 6)) * 4 + rowi;
    const bool live = frame < nframes;
    const size_t fr = min(frame, nframes - 1);
    const int H = p.H, nj = (H + 15) / 16;
    const float f0 = p.f0_hz[fr * p.S];
    const float inharm = fmaxf(p.inharm_coef[fr], 0.0f);
    float amp = apply_scale(p.scale, p.amplitudes[fr]);
    bool is_last = false;
    unsigned b_last = 0, tt = 0;
    if (p.shifts_last) {
        const unsigned row = (unsigned)(fr / (size_t)p.T), last_lo = (unsigned)(p.R / p.P) * (unsigned)(p.P - 1);
        tt = (unsigned)(fr - (size_t)row * p.T);
        is_last = live && (p.vmajor ? row >= last_lo : (row % (unsigned)p.P) == (unsigned)p.P - 1);
        b_last = p.vmajor ? row - last_lo : row / (unsigned)p.P;
    }
    const float* src = p.harmonic_distribution + fr * H;
    auto element = [&](int k, float& hd, float& shift, bool& above) {      // :37-44, :186
        const float m = (float)(k + 1);
        float g = m * m;
        g = g * inharm + 1.0f;
        g = sqrtf(g);
        shift = g - 1.0f;
        hd = apply_scale(p.scale, src[k]);
        above = (f0 * m) * g >= p.nyquist;
    };
    // pass 1: the sum the normalisation divides by (:194-198 before the cut, or :210-214 after it)
    float sum = 0.0f;
#pragma unroll 1
    for (int j = 0; j < nj; ++j) {
        const int k = sub + 16 * j;
        float hd = 0.0f, shift;
        bool above = false;
        if (k < H) element(k, hd, shift, above);
        if (p.normalize_after_nyquist_cut == 1 && p.normalize_below_nyquist && above) hd = 0.0f;
        if (k < H) sum += hd;
    }
    const float tot = row_sum(sum);
    const float den = tot == 0.0f ? 1e-7f : tot;                            // core.safe_divide
    if (p.normalize_below_nyquist) amp = amp * (f0 > p.min_frequency ? 1.0f : 0.0f);
    amp = amp / p.n_substrings;
    // pass 2: form, (normalise,) cut, (normalise,) store
    int last = 0;
#pragma unroll 1
    for (int j = 0; j < nj; ++j) {
        const int k = sub + 16 * j;
        if (k >= H) continue;
        float hd, shift;
        bool above;
        element(k, hd, shift, above);
        if (p.normalize_after_nyquist_cut == 0) hd = hd / den;
        if (p.normalize_below_nyquist && above) hd = 0.0f;
        if (p.normalize_after_nyquist_cut == 1) hd = hd / den;
        if (live) {
            p.hd_out[frame * H + k] = hd;
            if (p.shifts_out) p.shifts_out[frame * H + k] = shift;
            if (is_last) p.shifts_last[((size_t)b_last * p.T + tt) * H + k] = shift;
        }
        if (amp * hd != 0.0f) last = k + 1;
    }
    if (live && sub == 0) p.amp_out[frame] = amp;
    if (p.count_out) {
        last = row_max(last);
        if (live && sub == 0) p.count_out[frame] = last;
    }
}
__global__ void __launch_bounds__(256) inharmonic_controls_wide_kernel(const InharmParams p) { inharmonic_controls_wide_body(p); }

// (93 registers, five wavefronts per SIMD: held to 64 registers / eight wavefronts the compiler spreads the loads out
// between the uses and the kernel takes 370 us instead of 195; three or four passes per wavefront 215 / 250 us)
template <int NJ, int CTL_PASS>
__global__ void __launch_bounds__(256)
inharmonic_controls_kernel(const InharmParams p) {
    inharmonic_controls_body<NJ, CTL_PASS>(p);
}
// The same conditioning for the flags every model of the reference ships with (cut above Nyquist FIRST, then
// normalise: normalize_below_nyquist and normalize_after_nyquist_cut both set) and a whole number of 16-harmonic groups,
// with the scale function a template argument (round 4).  Same arithmetic per element and the same order of additions as
// inharmonic_controls_body -- the two agree bit for bit (tests/test_gpu_osc.py::test_lean_get_controls_*) -- but a third of the instructions:
//   * the body above re-decided the scale function with scalar branches at every element, kept hd[] / shift[] in
//     register tuples the compiler shuffled with ~25 v_mov_b64 per element, and wrapped every element in an EXEC branch;
//     here an element is straight-line code under one wave-uniform "is any of the four frames' groups alive" test;
//   * the inharmonicity factor sqrt(1 + B k^2) only DECIDES here (is the partial at or above Nyquist?) unless the frame
//     belongs to a segment's last voice, whose shifts the outputs dictionary keeps: the decision is taken from the
//     hardware's 1-ulp v_sqrt_f32 and redone with the correctly rounded square root (fifteen instructions) only for a
//     wavefront that holds a partial within 1e-6 of Nyquist -- the two can differ by 2^-22 at most; the shifts that are
//     stored come from the correctly rounded one, in a loop of their own that most wavefronts skip.
template <int NJ, int KIND>
__global__ void __launch_bounds__(256) inharmonic_controls_lean_kernel(const InharmParams p) {
    constexpr int CTL_PASS = 2;
    const int lane = threadIdx.x & 63, sub = lane & 15, rowi = lane >> 4;
    const size_t nframes = (size_t)p.R * p.T;
    const size_t wave0 = ((size_t)blockIdx.x * 4 + (size_t)wave_uniform(threadIdx.x >> 6)) * (4 * CTL_PASS);
    if (wave0 >= nframes) return;
    constexpr int H = 16 * NJ;
    const float nyq = p.nyquist, near = p.nyquist * 1e-6f;
    float raw_f0[CTL_PASS], raw_in[CTL_PASS], raw_amp[CTL_PASS];
#pragma unroll
    for (int u = 0; u < CTL_PASS; ++u) {
        const size_t fr = min(wave0 + 4 * u + rowi, nframes - 1);
        raw_f0[u] = p.f0_hz[fr * p.S];                                  // f0_hz[..., 0:1]  (:264)
        raw_in[u] = p.inharm_coef[fr];
        raw_amp[u] = p.amplitudes[fr];
    }
    unsigned row0 = 0, tt0 = 0, last_lo = 0;
    if (p.shifts_last) {
        row0 = (unsigned)wave0 / (unsigned)p.T;
        tt0 = (unsigned)wave0 - row0 * (unsigned)p.T;
        last_lo = (unsigned)(p.R / p.P) * (unsigned)(p.P - 1);          // voice major: first row of the last voice
    }
    unsigned dead[CTL_PASS];           // bit j: harmonics 16 j .. 16 j + 15 of the lane's frame are cut whole
    float raw_hd[CTL_PASS][NJ];
    const size_t wframes = min((size_t)(4 * CTL_PASS), nframes - wave0);
    const __amdgpu_buffer_rsrc_t window = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.harmonic_distribution + wave0 * H), 0, (int)(wframes * H * sizeof(float)), 0x00020000);
#pragma unroll
    for (int u = 0; u < CTL_PASS; ++u) {
        const size_t frame = wave0 + 4 * u + rowi;
        unsigned d = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (raw_f0[u] * (float)(16 * j + 1) >= nyq) d |= 1u << j;
        if (frame >= nframes) d = ~0u;
        dead[u] = d;
        // (buffer loads: a cut group's lanes point past the end of the wavefront's 8-frame window and get their 0.0
        // from the bounds check, without a memory access and without a branch -- all sixteen loads are in flight at once;
        // as `dead ? 0 : src[k]` each became an EXEC branch holding the load AND the wait for it)
        const unsigned at = (unsigned)((4 * u + rowi) * H + sub) * 4u;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            raw_hd[u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(window, ((d >> j) & 1) ? 0x40000000u : at + 64u * j, 0, 0));
    }
    const float subf = (float)(sub + 1);
#pragma unroll
    for (int u = 0; u < CTL_PASS; ++u) {
        const size_t frame = wave0 + 4 * u + rowi;
        const bool live = frame < nframes;
        const float f0 = raw_f0[u];
        const float inharm = fmaxf(raw_in[u], 0.0f);                    // :183
        float amp = scale_of<KIND>(p.scale, raw_amp[u]);                // :185
        // (twelve scalars behind constant pointers, not `float hd[NJ]`: as an array the values live in a register tuple
        // that the compiler shuffles with v_mov_b64 around every wave-uniform branch -- 69 moves at NJ = 8, 307 at 12)
        float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, h4 = 0.f, h5 = 0.f, h6 = 0.f, h7 = 0.f, h8 = 0.f, h9 = 0.f, h10 = 0.f, h11 = 0.f;
        float* const hd[12] = {&h0, &h1, &h2, &h3, &h4, &h5, &h6, &h7, &h8, &h9, &h10, &h11};
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool dj = (dead[u] >> j) & 1;
            if (!__all(dj)) {                                           // (wave-uniform)
                const float m = subf + (float)(16 * j);                 // linspace(1, H, H)
                float x = m * m;                       // tf.math.pow(int_multiplier, 2)        :37
                x = x * inharm + 1.0f;                 //                                        :38
                const float fm = f0 * m;
                float freq = fm * __builtin_amdgcn_sqrtf(x);            // f0_hz * int_multiplier * inharm_factor :42
                if (__any(fabsf(freq - nyq) <= near)) {
                    asm volatile("; a partial at Nyquist: the correctly rounded square root decides");    // (and keeps this a branch)
                    freq = fm * sqrtf(x);
                }
                const float v = scale_of<KIND>(p.scale, raw_hd[u][j]);                      // :186
                const float c = (dj || freq >= nyq) ? 0.0f : v;         // core.remove_above_nyquist :200-208
                *hd[j] = c;
                sum += c;
            }
        }
        amp = amp * (f0 > p.min_frequency ? 1.0f : 0.0f);
        {                                                               // :210-214
            const float tot = row_sum(sum);
            const float den = tot == 0.0f ? 1e-7f : tot;                // core.safe_divide
            const float rden = 1.0f / den;
            if (__all(den > 1e-30f && den < 1e30f)) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) *hd[j] = div_const(*hd[j], den, rden);
            } else {                                                    // (the IEEE division, whatever it gives)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (!((dead[u] >> j) & 1)) *hd[j] = *hd[j] / den;
            }
        }
        amp = amp / p.n_substrings;                                     // :269 (1.0 for InHarmonic)
        int last = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (amp * *hd[j] != 0.0f) last = sub + 16 * j + 1;
        // harmonic_shifts: every frame's when the caller keeps them all, else those of the segments' last voices
        bool is_last = false;
        unsigned row = row0, tt = tt0 + 4 * (unsigned)u + (unsigned)rowi;
        if (p.shifts_last) {
            while (tt >= (unsigned)p.T) {
                tt -= (unsigned)p.T;
                ++row;
            }
            is_last = live && (p.vmajor ? row >= last_lo : (row % (unsigned)p.P) == (unsigned)p.P - 1);
        }
        if (p.count_out) last = row_max(last);
        if (live) {
            float* dst = p.hd_out + frame * H + sub;
            if (p.hd_sparse) {                                          // (InharmParams::hd_sparse; wave-uniform)
                const int keep = is_last ? H : last;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (16 * j < keep) dst[16 * j] = *hd[j];
            } else {
#pragma unroll
                for (int j = 0; j < NJ; ++j) dst[16 * j] = *hd[j];
            }
            if (sub == 0) p.amp_out[frame] = amp;
        }
        if (p.count_out && live && sub == 0) p.count_out[frame] = last;  // bit 16 is added by frames_moved_kernel
        if (__any(live && (p.shifts_out != nullptr || is_last))) {
            const unsigned b = p.vmajor ? row - last_lo : row / (unsigned)p.P;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float m = subf + (float)(16 * j);
                float g = m * m;
                g = g * inharm + 1.0f;
                g = sqrtf(g);                          //                                        :39
                const float shift = g - 1.0f;          //                                        :44
                if (live && p.shifts_out) p.shifts_out[frame * H + sub + 16 * j] = shift;
                if (is_last) p.shifts_last[((size_t)b * p.T + tt) * H + sub + 16 * j] = shift;
            }
        }
    }
}

// Bit 16 of the per-frame info word: the frame's frequencies may differ from the previous frame's (some f0 sub-string
// or the clamped inharmonicity coefficient moved; never set on a row's first frame).  Equal inputs give equal harmonic
// frequencies, so a clear bit is a guarantee; the oscillator pre-pass finds its constant chunks with it.  A kernel of
// its own over the [R, T] scalars: inside the get_controls kernel the same test cost 0.05 ms.
__global__ void __launch_bounds__(256) frames_moved_kernel(const float* __restrict__ f0, const float* __restrict__ inh,
                                                         int* __restrict__ info, int R, int T, int S) {
    const size_t n = (size_t)R * T;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (size_t)gridDim.x * 256) {
        const int t = (int)(g % (size_t)T);
        if (t == 0) continue;
        bool moved = fmaxf(inh[g], 0.0f) != fmaxf(inh[g - 1], 0.0f);
        for (int s = 0; s < S; ++s) moved = moved || f0[g * S + s] != f0[(g - 1) * S + s];
        if (moved) info[g] |= 1 << 16;
    }
}

__global__ void __launch_bounds__(256) scale_bias_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       size_t n, float bias, ScaleFn s) {
    with_scale_kind(s.kind, [&](auto kind) {
        constexpr int KIND = decltype(kind)::value < 0 ? SCALE_NONE : decltype(kind)::value;
        for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (size_t)gridDim.x * 256)
            y[g] = scale_of<KIND>(s, x[g] + bias);
    });
}

// out = (((s0 + s1) + s2) + ...) elementwise: python `sum(signals.values())` of MultiAdd.
__global__ void __launch_bounds__(256) add_signals_kernel(const float* const* __restrict__ srcs, int nsrc,
                                                        float* __restrict__ out, size_t n4) {
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n4; g += (size_t)gridDim.x * 256) {
        float4 acc = reinterpret_cast<const float4*>(srcs[0])[g];
        for (int s = 1; s < nsrc; ++s) {
            const float4 v = reinterpret_cast<const float4*>(srcs[s])[g];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4*>(out)[g] = acc;
    }
}

// dry[b, n] = sum over voices p of (noise[b, p, n] + additive[b, p, n]), accumulated in the order of
// the polyphonic DAG: ((add + noise_p) + additive_p)   (polyphonic_dag.py:28-37)
__global__ void __launch_bounds__(256) polyphonic_mix_kernel(const float* __restrict__ additive,
                                                           const float* __restrict__ noise,
                                                           float* __restrict__ out, int B, int P,
                                                           int N, int out_stride, int voice_major,
                                                           float* __restrict__ out_prev) {
    const int n4 = N / 4;
    const size_t total = (size_t)B * n4;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int b = (int)(g / n4), i = (int)(g - (size_t)b * n4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int v = 0; v < P; ++v) {
            // the running mix the last `add` node receives as its first operand (its controls' signal_0)
            if (out_prev && v == P - 1) reinterpret_cast<float4*>(out_prev + (size_t)b * N)[i] = acc;
            const size_t off = (voice_major ? (size_t)v * B + b : (size_t)b * P + v) * n4 + i;
            if (noise) {
                const float4 z = reinterpret_cast<const float4*>(noise)[off];
                if (v == 0) acc = z;
                else { acc.x += z.x; acc.y += z.y; acc.z += z.z; acc.w += z.w; }
            }
            const float4 a = reinterpret_cast<const float4*>(additive)[off];
            if (v == 0 && !noise) acc = a;
            else { acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
        }
        reinterpret_cast<float4*>(out + (size_t)b * out_stride)[i] = acc;
    }
}

// Every node of the add chain of default_model.py:56-74, with all its intermediate signals: sub[b, v] = noise[b, v] +
// additive[b, v] (`sub_add_v`; for v = 0 this is `add_0` itself), run[b, 0] = sub[b, 0], run[b, v] = run[b, v - 1] + sub[b, v]
// (`add_v`).  One pass over the stems instead of 2 P - 1 launches.
__global__ void __launch_bounds__(256) add_chain_paired_kernel(const float* __restrict__ additive,
                                                             const float* __restrict__ noise, float* __restrict__ sub,
                                                             float* __restrict__ run, int B, int P, int N, int voice_major) {
    const int n4 = N / 4;
    const size_t total = (size_t)B * n4;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int b = (int)(g / n4), i = (int)(g - (size_t)b * n4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int v = 0; v < P; ++v) {
            const size_t off = (voice_major ? (size_t)v * B + b : (size_t)b * P + v) * n4 + i;
            const float4 z = reinterpret_cast<const float4*>(noise)[off], a = reinterpret_cast<const float4*>(additive)[off];
            const float4 s = make_float4(z.x + a.x, z.y + a.y, z.z + a.z, z.w + a.w);
            if (v == 0) acc = s;
            else { acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w; }
            const size_t o = ((size_t)b * P + v) * n4 + i;
            reinterpret_cast<float4*>(sub)[o] = s;
            reinterpret_cast<float4*>(run)[o] = acc;
        }
    }
}

// out[b, n] = sum_v a[b, v, n] (PA rows) + sum_v z[b, v, n] (PZ rows): the add chain when one operand is
// already a per-segment mix
// tail_z / tail_a / out_prev (all or none): out_prev = the sum above, out = (out_prev + tail_z) + tail_a -- the last step
// of the add chain of polyphonic_dag.py:34-37 with its three operands kept apart; with out_sub the last step as
// default_model.py:68-74 writes it: out_sub = tail_z + tail_a, out = out_prev + out_sub
__global__ void __launch_bounds__(256) mix_voices_kernel(const float* __restrict__ a, int PA,
                                                       const float* __restrict__ z, int PZ,
                                                       float* __restrict__ out, int B, int N, int out_stride,
                                                       int voice_major, const float* __restrict__ tail_z,
                                                       const float* __restrict__ tail_a, float* __restrict__ out_prev,
                                                       float* __restrict__ out_sub) {
    const int n4 = N / 4;
    const size_t total = (size_t)B * n4;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int b = (int)(g / n4), i = (int)(g - (size_t)b * n4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int v = 0; v < PZ; ++v) {
            const float4 t = reinterpret_cast<const float4*>(z)[(voice_major ? (size_t)v * B + b : (size_t)b * PZ + v) * n4 + i];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        for (int v = 0; v < PA; ++v) {
            const float4 t = reinterpret_cast<const float4*>(a)[(voice_major ? (size_t)v * B + b : (size_t)b * PA + v) * n4 + i];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        if (out_prev) {
            reinterpret_cast<float4*>(out_prev + (size_t)b * N)[i] = acc;
            const float4 tz = reinterpret_cast<const float4*>(tail_z + (size_t)b * N)[i];
            const float4 ta = reinterpret_cast<const float4*>(tail_a + (size_t)b * N)[i];
            if (out_sub) {        // default_model.py:68-74: sub_add_i = noise + additive, add_i = add_{i-1} + sub_add_i
                const float4 sb = make_float4(tz.x + ta.x, tz.y + ta.y, tz.z + ta.z, tz.w + ta.w);
                reinterpret_cast<float4*>(out_sub + (size_t)b * N)[i] = sb;
                acc.x = acc.x + sb.x; acc.y = acc.y + sb.y; acc.z = acc.z + sb.z; acc.w = acc.w + sb.w;
            } else {
                acc.x = (acc.x + tz.x) + ta.x; acc.y = (acc.y + tz.y) + ta.y;
                acc.z = (acc.z + tz.z) + ta.z; acc.w = (acc.w + tz.w) + ta.w;
            }
        }
        reinterpret_cast<float4*>(out + (size_t)b * out_stride)[i] = acc;
    }
}

static unsigned stream_grid(size_t total) {
    size_t blocks = (total + 255) / 256;
    const size_t cap = 256 * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace ddspp

using namespace ddspp;

extern "C" {

// InHarmonic.get_controls / MultiInharmonic.get_controls  -- inharm_synth.py:167-219, :254-270.
// scale_kind: 0 none, 1 core.exp_sigmoid, 2 exp_tanh; (exponent, max_value, threshold, gain) are the
// keyword defaults of those functions unless the caller overrides them.
static int inharmonic_controls_impl(const float* amplitudes, const float* harmonic_distribution,
                              const float* inharm_coef, const float* f0_hz, float* amplitudes_out,
                              float* harmonic_distribution_out, float* harmonic_shifts_out, int* audible_out,
                              int R, int T, int H, int S, float sample_rate, float min_frequency, int scale_kind,
                              float exponent, float max_value, float threshold, float gain,
                              int normalize_after_nyquist_cut, int normalize_below_nyquist,
                              float* shifts_last_out, int n_voices, int voice_major,
                              hipStream_t stream, int hd_sparse = 0) {
    DDSPP_REQUIRE(amplitudes && harmonic_distribution && inharm_coef && f0_hz && amplitudes_out &&
                      harmonic_distribution_out,
                  "inharmonic_controls: null buffer");
    DDSPP_REQUIRE(R > 0 && T > 0 && H > 0 && S > 0, "inharmonic_controls: bad dims");
    DDSPP_REQUIRE(H <= 512, "inharmonic_controls: n_harmonics=%d exceeds 512", H);
    DDSPP_REQUIRE(scale_kind >= 0 && scale_kind <= 2, "inharmonic_controls: unknown scale_fn %d", scale_kind);
    DDSPP_REQUIRE(normalize_after_nyquist_cut >= 0 && normalize_after_nyquist_cut <= 2,
                  "inharmonic_controls: normalize_after_nyquist_cut=%d (0 before the cut, 1 after it, 2 never)", normalize_after_nyquist_cut);
    const bool norm_after = normalize_after_nyquist_cut == 1;
    InharmParams p{};
    p.amplitudes = amplitudes; p.harmonic_distribution = harmonic_distribution;
    p.inharm_coef = inharm_coef; p.f0_hz = f0_hz;
    p.amp_out = amplitudes_out; p.hd_out = harmonic_distribution_out; p.shifts_out = harmonic_shifts_out;
    p.count_out = audible_out;
    DDSPP_REQUIRE(!shifts_last_out || (n_voices >= 1 && R % n_voices == 0 && (long long)R * T < (1ll << 31)),
                  "inharmonic_controls: %d rows are not a whole number of %d-voice segments (or too many frames)", R, n_voices);
    p.shifts_last = shifts_last_out; p.P = n_voices; p.vmajor = voice_major;
    p.R = R; p.T = T; p.H = H; p.S = S;
    p.nyquist = sample_rate / 2.0f; p.min_frequency = min_frequency; p.n_substrings = (float)S;
    p.normalize_after_nyquist_cut = normalize_after_nyquist_cut;
    p.normalize_below_nyquist = normalize_below_nyquist;
    p.scale = ScaleFn{scale_kind, logf(exponent), max_value, threshold, gain};
    const size_t frames = (size_t)R * T;
    const int nj = (H + 15) / 16;
    const bool two_pass = nj <= 8 || (nj == 12 && norm_after && normalize_below_nyquist && H % 16 == 0 &&
                                      !ddspp_option_literal("DDSPP_CONTROLS_GENERIC", 0));          // (the lean kernel: always two)
    const size_t per_wg = (size_t)4 * 4 * (two_pass ? 2 : 1);           // four wavefronts of 4 * CTL_PASS frames
    const dim3 grid((unsigned)((frames + per_wg - 1) / per_wg)), block(256);
    // every shipped model: 48 / 64 / 96 / 128 / 192 harmonics (8, 16, 16 / 24, 24 / 48, 32 kHz) with the default flags
    const bool lean = norm_after && normalize_below_nyquist && H % 16 == 0 &&
                      (nj == 3 || nj == 4 || nj == 6 || nj == 8 || nj == 12) && !ddspp_option_literal("DDSPP_CONTROLS_GENERIC", 0);
    // (the sparse store is the lean kernel's; any other shape or flag set writes the whole tensor, which is always valid)
    p.hd_sparse = (hd_sparse && lean && audible_out && !ddspp_option_literal("DDSPP_CONTROLS_DENSE_HD", 0)) ? 1 : 0;
#define DDSPP_LEAN(NJ)                                                                                              \
    do {                                                                                                            \
        if (scale_kind == SCALE_EXP_SIGMOID)                                                                        \
            hipLaunchKernelGGL((inharmonic_controls_lean_kernel<NJ, SCALE_EXP_SIGMOID>), grid, block, 0, stream, p); \
        else if (scale_kind == SCALE_EXP_TANH)                                                                      \
            hipLaunchKernelGGL((inharmonic_controls_lean_kernel<NJ, SCALE_EXP_TANH>), grid, block, 0, stream, p);    \
        else                                                                                                        \
            hipLaunchKernelGGL((inharmonic_controls_lean_kernel<NJ, SCALE_NONE>), grid, block, 0, stream, p);        \
    } while (0)
    if (lean && nj == 8) DDSPP_LEAN(8);
    else if (lean && nj == 6) DDSPP_LEAN(6);
    else if (lean && nj == 4) DDSPP_LEAN(4);
    else if (lean && nj == 3) DDSPP_LEAN(3);
    else if (lean && nj == 12) DDSPP_LEAN(12);
    else if (nj <= 4) hipLaunchKernelGGL((inharmonic_controls_kernel<4, 2>), grid, block, 0, stream, p);
    else if (nj <= 6) hipLaunchKernelGGL((inharmonic_controls_kernel<6, 2>), grid, block, 0, stream, p);
    else if (nj <= 8) hipLaunchKernelGGL((inharmonic_controls_kernel<8, 2>), grid, block, 0, stream, p);
    else if (nj <= 12) hipLaunchKernelGGL((inharmonic_controls_kernel<12, 1>), grid, block, 0, stream, p);
    else if (nj <= 16) hipLaunchKernelGGL((inharmonic_controls_kernel<16, 1>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL(inharmonic_controls_wide_kernel, grid, block, 0, stream, p);      // 257 .. 512 harmonics
    if (audible_out)
        hipLaunchKernelGGL(frames_moved_kernel, dim3(stream_grid(frames)), dim3(256), 0, stream, f0_hz, inharm_coef,
                           audible_out, R, T, S);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

int ddspp_inharmonic_controls(const float* amplitudes, const float* harmonic_distribution,
                              const float* inharm_coef, const float* f0_hz, float* amplitudes_out,
                              float* harmonic_distribution_out, float* harmonic_shifts_out, int* audible_out,
                              int R, int T, int H, int S, float sample_rate, float min_frequency, int scale_kind,
                              float exponent, float max_value, float threshold, float gain,
                              int normalize_after_nyquist_cut, int normalize_below_nyquist,
                              hipStream_t stream) {
    return inharmonic_controls_impl(amplitudes, harmonic_distribution, inharm_coef, f0_hz, amplitudes_out,
                                    harmonic_distribution_out, harmonic_shifts_out, audible_out, R, T, H, S, sample_rate,
                                    min_frequency, scale_kind, exponent, max_value, threshold, gain,
                                    normalize_after_nyquist_cut, normalize_below_nyquist, nullptr, 1, 0, stream);
}

// The same over the R = n_segments * n_voices rows of a polyphonic group (segment major, or voice major), writing
// harmonic_shifts only for every segment's LAST voice (shifts_last_out [R / n_voices, T, H]): the compacted oscillator
// bank forms the shifts of all voices itself, the outputs dictionary of the reference's DAG keeps the last voice's.
int ddspp_inharmonic_controls_group(const float* amplitudes, const float* harmonic_distribution,
                                    const float* inharm_coef, const float* f0_hz, float* amplitudes_out,
                                    float* harmonic_distribution_out, float* shifts_last_out, int* audible_out,
                                    int R, int T, int H, int S, int n_voices, int voice_major, float sample_rate,
                                    float min_frequency, int scale_kind, float exponent, float max_value,
                                    float threshold, float gain, int normalize_after_nyquist_cut,
                                    int normalize_below_nyquist, hipStream_t stream) {
    return inharmonic_controls_impl(amplitudes, harmonic_distribution, inharm_coef, f0_hz, amplitudes_out,
                                    harmonic_distribution_out, nullptr, audible_out, R, T, H, S, sample_rate,
                                    min_frequency, scale_kind, exponent, max_value, threshold, gain,
                                    normalize_after_nyquist_cut, normalize_below_nyquist, shifts_last_out, n_voices,
                                    voice_major, stream);
}

// ddspp_inharmonic_controls_group for a caller whose only reader of harmonic_distribution_out is the compacted oscillator
// bank (ddspp_polyphonic_additive / ddspp_polyphonic_stems with `audible` = audible_out, required here): a frame's row is
// written only below its audible count, rounded up to whole groups of 16 harmonics -- every value left out would have been
// multiplied by an amplitude to exactly zero, and the bank takes it as zero from the count -- except for the rows of every
// segment's last voice (when shifts_last_out is given), which are written whole for the outputs dictionary.  The rest of
// the buffer keeps what it held: it is NOT a valid harmonic_distribution for any other reader.  Saves two thirds of the
// tensor's write at a piano's note mix (config 3: 0.42 -> 0.15 GB).  DDSPP_CONTROLS_DENSE_HD=1: write everything (A/B).
int ddspp_inharmonic_controls_sparse(const float* amplitudes, const float* harmonic_distribution,
                                     const float* inharm_coef, const float* f0_hz, float* amplitudes_out,
                                     float* harmonic_distribution_out, float* shifts_last_out, int* audible_out,
                                     int R, int T, int H, int S, int n_voices, int voice_major, float sample_rate,
                                     float min_frequency, int scale_kind, float exponent, float max_value,
                                     float threshold, float gain, int normalize_after_nyquist_cut,
                                     int normalize_below_nyquist, hipStream_t stream) {
    DDSPP_REQUIRE(audible_out, "inharmonic_controls_sparse: audible_out is what tells the reader where a row ends");
    return inharmonic_controls_impl(amplitudes, harmonic_distribution, inharm_coef, f0_hz, amplitudes_out,
                                    harmonic_distribution_out, nullptr, audible_out, R, T, H, S, sample_rate,
                                    min_frequency, scale_kind, exponent, max_value, threshold, gain,
                                    normalize_after_nyquist_cut, normalize_below_nyquist, shifts_last_out, n_voices,
                                    voice_major, stream, 1);
}

// SurrogateAdditive.get_controls' decay factors -- ddsp_piano/modules/surrogate_synth.py:163-171:
// decays_out = where(inharmonic_freq >= sample_rate / 2, 1, clip(decays, 1e-5, 1)), inharmonic_freq = (f0 k) g with
// g = sqrt(k^2 max(inharm_coef, 0) + 1) formed as get_inharmonic_freq does (inharm_synth.py:37-42, osc_common.h).
__global__ void __launch_bounds__(256) surrogate_decays_kernel(const float* __restrict__ decays, const float* __restrict__ inharm_coef,
                                                             const float* __restrict__ f0_hz, float* __restrict__ out, size_t frames,
                                                             int H, float nyquist) {
    const size_t total = frames * (size_t)H;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const size_t fr = g / (size_t)H;
        const float m = (float)(int)(g - fr * (size_t)H + 1);
        const float inharm = fmaxf(inharm_coef[fr], 0.0f);
        float q = m * m;
        q = q * inharm + 1.0f;
        q = sqrtf(q);
        const float freq = (f0_hz[fr] * m) * q;
        const float d = fmaxf(fminf(decays[g], 1.0f), 1e-5f);                     // :165-166 (minimum, then maximum)
        out[g] = freq >= nyquist ? 1.0f : d;
    }
}

int ddspp_surrogate_decays(const float* decays, const float* inharm_coef, const float* f0_hz, float* decays_out, int R, int T,
                           int H, float sample_rate, hipStream_t stream) {
    DDSPP_REQUIRE(decays && inharm_coef && f0_hz && decays_out, "surrogate_decays: null buffer");
    DDSPP_REQUIRE(R > 0 && T > 0 && H > 0, "surrogate_decays: bad dims");
    const size_t frames = (size_t)R * T;
    hipLaunchKernelGGL(surrogate_decays_kernel, dim3(stream_grid(frames * (size_t)H)), dim3(256), 0, stream, decays, inharm_coef,
                       f0_hz, decays_out, frames, H, sample_rate / 2.0f);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// ddsp.synths.FilteredNoise.get_controls: magnitudes = scale_fn(magnitudes + initial_bias)
int ddspp_scale_bias(const float* x, float* y, size_t n, float bias, int scale_kind, float exponent,
                     float max_value, float threshold, float gain, hipStream_t stream) {
    DDSPP_REQUIRE(x && y, "scale_bias: null buffer");
    DDSPP_REQUIRE(scale_kind >= 0 && scale_kind <= 2, "scale_bias: unknown scale_fn %d", scale_kind);
    if (n == 0) return DDSPP_OK;
    ScaleFn s{scale_kind, logf(exponent), max_value, threshold, gain};
    hipLaunchKernelGGL(scale_bias_kernel, dim3(stream_grid(n)), dim3(256), 0, stream, x, y, n, bias, s);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// MultiAdd.get_signal / ddsp.processors.Add.get_signal  -- inharm_synth.py:308-309.
// `srcs` is a DEVICE array of nsrc device pointers, each to n floats (n % 4 == 0, 16-byte aligned).
int ddspp_add_signals(const float* const* srcs, int nsrc, float* out, size_t n, hipStream_t stream) {
    DDSPP_REQUIRE(srcs && out && nsrc >= 1, "add_signals: bad arguments");
    DDSPP_REQUIRE(n % 4 == 0, "add_signals: n must be a multiple of 4");
    if (n == 0) return DDSPP_OK;
    hipLaunchKernelGGL(add_signals_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, stream, srcs, nsrc, out,
                       n / 4);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// The `add` chain of polyphonic_dag.py:28-37 for all voices at once: additive/noise are
// [B, P, N] (voice_major = 0) or [P, B, N] (voice_major = 1, the layout the reference's Parallelizer
// leaves the merged controls in, sub_modules.py:573-592); out rows are written with a stride (so the
// dry mix can land in a zero-padded FFT buffer of the reverb).  noise may be null (dry additive only).
int ddspp_polyphonic_mix(const float* additive, const float* noise, float* out, float* out_prev, int B, int P, int N,
                         int out_stride, int voice_major, hipStream_t stream) {
    DDSPP_REQUIRE(additive && out, "polyphonic_mix: null buffer");
    DDSPP_REQUIRE(B > 0 && P > 0 && N > 0 && N % 4 == 0 && out_stride % 4 == 0 && out_stride >= N,
                  "polyphonic_mix: bad dims");
    hipLaunchKernelGGL(polyphonic_mix_kernel, dim3(stream_grid((size_t)B * (N / 4))), dim3(256), 0, stream,
                       additive, noise, out, B, P, N, out_stride, voice_major, out_prev);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// The whole add chain of default_model.py:56-74 with every node's signal: additive / noise [B,P,N] (or [P,B,N]) ->
// sub [B,P,N] (`sub_add_v`; sub[:, 0] = `add_0`), run [B,P,N] (`add_v`), both segment major.
int ddspp_add_chain_paired(const float* additive, const float* noise, float* sub, float* run, int B, int P, int N,
                           int voice_major, hipStream_t stream) {
    DDSPP_REQUIRE(additive && noise && sub && run, "add_chain_paired: null buffer");
    DDSPP_REQUIRE(B > 0 && P > 0 && N > 0 && N % 4 == 0, "add_chain_paired: bad dims");
    hipLaunchKernelGGL(add_chain_paired_kernel, dim3(stream_grid((size_t)B * (N / 4))), dim3(256), 0, stream, additive, noise,
                       sub, run, B, P, N, voice_major);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// out[b] = sum of PA rows of a[b] + sum of PZ rows of z[b] (either may be NULL with its count 0).
int ddspp_mix_voices(const float* a, int PA, const float* z, int PZ, float* out, int B, int N, int out_stride,
                     int voice_major, hipStream_t stream) {
    DDSPP_REQUIRE(out && (a || PA == 0) && (z || PZ == 0) && PA >= 0 && PZ >= 0 && PA + PZ > 0,
                  "mix_voices: bad arguments");
    DDSPP_REQUIRE(B > 0 && N > 0 && N % 4 == 0 && out_stride % 4 == 0 && out_stride >= N, "mix_voices: bad dims");
    hipLaunchKernelGGL(mix_voices_kernel, dim3(stream_grid((size_t)B * (N / 4))), dim3(256), 0, stream, a, PA, z, PZ,
                       out, B, N, out_stride, voice_major, (const float*)nullptr, (const float*)nullptr, (float*)nullptr,
                       (float*)nullptr);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// The end of the add chain with the last voice kept apart (what the reference's outputs dictionary holds for the
// re-used processors, polyphonic_dag.py:34-37): prev[b] = sum of the PA rows a[b, :] + the PZ rows z[b, :] (the other
// voices), dry[b] = (prev[b] + noise_last[b]) + additive_last[b].  All outputs [B, N].
int ddspp_mix_last_voice(const float* a, int PA, const float* z, int PZ, const float* noise_last,
                         const float* additive_last, float* prev, float* dry, int B, int N, int voice_major,
                         hipStream_t stream) {
    DDSPP_REQUIRE(prev && dry && noise_last && additive_last && (a || PA == 0) && (z || PZ == 0) && PA >= 0 && PZ >= 0,
                  "mix_last_voice: bad arguments");
    DDSPP_REQUIRE(B > 0 && N > 0 && N % 4 == 0, "mix_last_voice: bad dims");
    hipLaunchKernelGGL(mix_voices_kernel, dim3(stream_grid((size_t)B * (N / 4))), dim3(256), 0, stream, a, PA, z, PZ,
                       dry, B, N, N, voice_major, noise_last, additive_last, prev, (float*)nullptr);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// The same for the node list of default_model.py:44-80, whose last voice is added as a pair: sub[b] = noise_last[b] +
// additive_last[b] (`sub_add_{P-1}`), dry[b] = prev[b] + sub[b] (`add_{P-1}`); prev = `add_{P-2}`.  All [B, N].
int ddspp_mix_last_voice_paired(const float* a, int PA, const float* z, int PZ, const float* noise_last,
                                const float* additive_last, float* prev, float* sub, float* dry, int B, int N,
                                int voice_major, hipStream_t stream) {
    DDSPP_REQUIRE(prev && sub && dry && noise_last && additive_last && (a || PA == 0) && (z || PZ == 0) && PA >= 0 && PZ >= 0,
                  "mix_last_voice_paired: bad arguments");
    DDSPP_REQUIRE(B > 0 && N > 0 && N % 4 == 0, "mix_last_voice_paired: bad dims");
    hipLaunchKernelGGL(mix_voices_kernel, dim3(stream_grid((size_t)B * (N / 4))), dim3(256), 0, stream, a, PA, z, PZ,
                       dry, B, N, N, voice_major, noise_last, additive_last, prev, sub);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

}  // extern "C"
