// Host builders of the small tables the kernels of libddspp take (C-ABI, include/ddspp.h): a caller that binds the
// library without the Python layer gets everything from here.  They restate, with the float32 /
// float64 arithmetic of the TF kernels:
//   tf.signal.hann_window                                   (window_ops._raised_cosine_window)
//   tf.compat.v1.image.resize(BILINEAR, align_corners=False) source rows / weights  (resize_bilinear CPU kernel), as
//       reached through ddsp.core.resample(method='linear')                          -- call site inharm_synth.py:117
//   ddsp.core.frequency_impulse_response = irfft + apply_window_to_impulse_response  -- filtered_noise_synth.py:41-42
// Outputs are HOST buffers; the caller copies them to the device (it owns every buffer).
// ddsp_piano_amd/core.py builds the same tables with numpy (they agree to the last float32 bit except where a libm
// cosine differs by an ulp; tests/test_cabi.py compares them).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "ddspp_common.h"

namespace {

const double kPi = 3.14159265358979323846;

// tf.signal.hann_window(n, periodic): 0.5 - 0.5 * cos(2 pi i / (n + periodic * even - 1)), every op in float32
void hann_f32(int n, int periodic, float* w) {
    if (n == 1) {
        w[0] = 1.0f;
        return;
    }
    const int even = 1 - n % 2;
    const float denom = (float)(n + (periodic ? 1 : 0) * even - 1);
    const float two_pi = DDSPP_TWO_PI_F32;
    for (int i = 0; i < n; ++i) {
        const float arg = (two_pi * (float)i) / denom;
        w[i] = 0.5f - 0.5f * cosf(arg);
    }
}

// ddsp.core.apply_window_to_impulse_response(causal=False) on one zero-phase row `ir` (double, length ir_size);
// crop_rule 0: ddsp 3.7.0 as recalled, 1: 'centred' (DESIGN.md section 2).  Returns the output length.
int apply_window_row(const double* ir, int ir_size, int window_size, int crop_rule, double* out) {
    if (window_size <= 0 || window_size > ir_size) window_size = ir_size;
    std::vector<float> wf(window_size);
    hann_f32(window_size, 1, wf.data());
    const int padding = ir_size - window_size;
    std::vector<double> win(ir_size, 0.0), prod(ir_size);
    if (padding > 0) {
        const int half_idx = crop_rule == 1 ? window_size / 2 : (window_size + 1) / 2;
        // window = concat(window[half_idx:], zeros(padding), window[:half_idx])
        int o = 0;
        for (int i = half_idx; i < window_size; ++i) win[o++] = wf[i];
        o += padding;
        for (int i = 0; i < half_idx; ++i) win[o++] = wf[i];
        for (int i = 0; i < ir_size; ++i) prod[i] = win[i] * ir[i];
        int first_half_start, second_half_end;
        if (crop_rule == 1) {
            first_half_start = ir_size - half_idx;
            second_half_end = window_size - half_idx;
        } else {
            first_half_start = (ir_size - (half_idx - 1)) + 1;
            second_half_end = half_idx + 1;
        }
        int n = 0;
        for (int i = first_half_start; i < ir_size; ++i) out[n++] = prod[i];
        for (int i = 0; i < second_half_end; ++i) out[n++] = prod[i];
        return n;
    }
    // fftshift of the window, multiply, fftshift of the product
    const int sh = ir_size / 2;                                   // np.fft.fftshift: roll by n // 2
    for (int i = 0; i < ir_size; ++i) win[(i + sh) % ir_size] = wf[i];
    for (int i = 0; i < ir_size; ++i) prod[i] = win[i] * ir[i];
    for (int i = 0; i < ir_size; ++i) out[(i + sh) % ir_size] = prod[i];
    return ir_size;
}

}  // namespace

extern "C" {

int ddspp_hann_window_host(int n, float* window) {
    DDSPP_REQUIRE(n >= 1 && window, "hann_window_host: bad arguments");
    hann_f32(n, 1, window);
    return DDSPP_OK;
}

// rule 0: TF1 legacy bilinear (pos = n * T/N); rule 1: half-pixel centres.
// aligned (may be NULL): 1 when N % T == 0 and lo[n] == n / (N / T) for every n (the fused oscillator path's condition).
int ddspp_resample_tables_host(int T, int N, int rule, int* lo, int* hi, float* w, int* aligned) {
    DDSPP_REQUIRE(T >= 1 && N >= 1 && lo && hi && w, "resample_tables_host: bad arguments");
    DDSPP_REQUIRE(rule == 0 || rule == 1, "resample_tables_host: unknown rule %d", rule);
    const float scale = (float)T / (float)N;
    bool ok = (N % T == 0);
    const int u = ok ? N / T : 1;
    for (int n = 0; n < N; ++n) {
        const float pos = rule == 1 ? ((float)n + 0.5f) * scale - 0.5f : (float)n * scale;
        const float fl = floorf(pos);
        int l = (int)fl, h = (int)ceilf(pos);
        if (l < 0) l = 0;
        if (l > T - 1) l = T - 1;       // (TF does not: the last sample of a file past 131 072 frames would read a row past the tensor)
        if (h < 0) h = 0;
        if (h > T - 1) h = T - 1;
        lo[n] = l;
        hi[n] = h;
        w[n] = pos - fl;
        if (ok && l != n / u) ok = false;
    }
    if (aligned) *aligned = ok ? 1 : 0;
    return DDSPP_OK;
}

// Interpolation weights of samples first_sample .. first_sample + n - 1 of a signal upsampled by N / T (a streamed
// piece, ddspp_polyphonic_additive with phase_state_in): the resize kernel multiplies float32(n) by the float32 scale
// at the ABSOLUTE n, and the rounding of that product depends on its magnitude -- a piece that wants the one-call
// render's numbers takes them from here.  first_sample = 0, n = N gives w of ddspp_resample_tables_host.
int ddspp_linear_weights_host(int T, int N, int rule, long long first_sample, int n, float* w) {
    DDSPP_REQUIRE(T >= 1 && N >= 1 && n >= 1 && first_sample >= 0 && w, "linear_weights_host: bad arguments");
    DDSPP_REQUIRE(rule == 0 || rule == 1, "linear_weights_host: unknown rule %d", rule);
    const float scale = (float)T / (float)N;
    for (int i = 0; i < n; ++i) {
        const float x = (float)(first_sample + i);
        const float pos = rule == 1 ? (x + 0.5f) * scale - 0.5f : x * scale;
        w[i] = pos - floorf(pos);
    }
    return DDSPP_OK;
}

// `wlin` as the frame-walking kernels take it (ddspp_harmonic_synthesis, ddspp_polyphonic_additive,
// ddspp_oscillator_phase_state): those kernels walk frame t = n / U and interpolate between rows t and t + 1, which IS the
// resize kernel's (lo, hi) pair as long as floor(float32(n) * float32(T / N)) == n / U.  Far into a long file (frame
// 131 073 at hop 96: what synthesize_midi_file.py renders for a piece of more than 8.7 minutes) the product rounds UP to
// the next whole frame for the last sample(s) of a frame: lo = hi = t + 1, weight 0 -- the sample takes x[t + 1] itself.
// Such samples get the mark 1.0 (a fractional part never is 1; WALK_NEXT_ROW in osc_common.h) and the kernels substitute
// x1 exactly.  *walkable = 0 when a sample needs anything else (a row below t, a row above with a non-zero weight, more
// marked samples than the 8 of a frame's last block, N not a multiple of T): the caller then takes the three-operator
// route with the full tables of ddspp_resample_tables_host.
int ddspp_walk_weights_host(int T, int N, int rule, long long first_sample, int n, float* w, int* walkable) {
    DDSPP_REQUIRE(T >= 1 && N >= 1 && n >= 1 && first_sample >= 0 && w, "walk_weights_host: bad arguments");
    DDSPP_REQUIRE(rule == 0 || rule == 1, "walk_weights_host: unknown rule %d", rule);
    const float scale = (float)T / (float)N;
    bool ok = (N % T == 0);
    const long long u = ok ? N / T : 1;
    for (int i = 0; i < n; ++i) {
        const long long idx = first_sample + i;
        const float x = (float)idx;
        const float pos = rule == 1 ? (x + 0.5f) * scale - 0.5f : x * scale;
        const float fl = floorf(pos);
        float wi = pos - fl;
        const long long t = idx / u, lo = (long long)fl < 0 ? 0 : (long long)fl;
        if (ok && lo != t) {
            // (the last frame of a whole signal clamps: rows T - 1, T - 1 either way -- callers pass pieces of one signal
            // whose controls hold a look-ahead frame, so t + 1 exists wherever it is asked for)
            if (lo == t + 1 && wi == 0.0f && idx % u >= u - 8) wi = 1.0f;
            else ok = false;
        }
        w[i] = wi;
    }
    if (walkable) *walkable = ok ? 1 : 0;
    return DDSPP_OK;
}

// Length of the FIRs frequency_impulse_response(magnitudes[..., K], window_size) returns, and the number of even/odd
// table rows NJ (0 when the even/odd design kernels do not take this shape: K not in {32, 64, 96, 128} or a cropped window).
int ddspp_fir_tables_shape(int K, int window_size, int* Lw, int* NJ) {
    DDSPP_REQUIRE(K >= 2 && Lw, "fir_tables_shape: bad arguments");
    const int ir_size = 2 * (K - 1);
    *Lw = (window_size <= 0 || window_size > ir_size) ? ir_size : window_size;
    if (NJ) {
        const bool eo = (K == 32 || K == 64 || K == 96 || K == 128) && !(window_size > 0 && window_size < ir_size);
        *NJ = eo ? (K - 1) / 2 + 1 : 0;
    }
    return DDSPP_OK;
}

// M[K, Lw] with frequency_impulse_response(mag) == mag @ M (real inverse DFT x window, shifted): the table of
// ddspp_fir_from_magnitudes.  uniq / mirror (int32[Lw] each, may be NULL) + n_uniq: the taps to evaluate and the tap
// each result is mirrored onto (-1: none); the windowed zero-phase response is even about its centre tap.
int ddspp_fir_matrix_host(int K, int window_size, int crop_rule, float* M, int* uniq, int* mirror, int* n_uniq) {
    DDSPP_REQUIRE(K >= 2 && M, "fir_matrix_host: bad arguments");
    DDSPP_REQUIRE(crop_rule == 0 || crop_rule == 1, "fir_matrix_host: unknown crop rule %d", crop_rule);
    const int ir_size = 2 * (K - 1);
    int lw = 0;
    std::vector<double> basis(ir_size), row(ir_size);
    std::vector<double> md;                                        // [K, lw] in double, for the symmetry check
    for (int k = 0; k < K; ++k) {
        const double coef = (k == 0 || k == K - 1) ? 1.0 : 2.0;
        for (int j = 0; j < ir_size; ++j) basis[j] = coef * cos(2.0 * kPi * (double)k * (double)j / ir_size) / ir_size;
        lw = apply_window_row(basis.data(), ir_size, window_size, crop_rule, row.data());
        if (k == 0) md.resize((size_t)K * lw);
        for (int i = 0; i < lw; ++i) {
            md[(size_t)k * lw + i] = (double)(float)row[i];
            M[(size_t)k * lw + i] = (float)row[i];
        }
    }
    if (!uniq || !mirror || !n_uniq) return DDSPP_OK;
    double mx = 0.0;
    for (double v : md) mx = fmax(mx, fabs(v));
    const double tol = 1e-6 * mx;
    const int cands[2] = {lw / 2, (lw - 1) / 2};
    for (int ci = 0; ci < 2; ++ci) {
        const int c = cands[ci];
        bool ok = true;
        for (int d = 1; d < lw && ok; ++d) {
            const int l = c - d, h = c + d;
            if (l < 0 && h >= lw) break;
            if (l >= 0 && h < lw)
                for (int k = 0; k < K; ++k)
                    if (fabs(md[(size_t)k * lw + l] - md[(size_t)k * lw + h]) > tol) {
                        ok = false;
                        break;
                    }
        }
        if (ok) {
            int n = 0;
            for (int i = 0; i < lw; ++i) {
                const int j = 2 * c - i;
                if (i >= c || j >= lw) {               // keep the upper half, and lower taps with no partner
                    uniq[n] = i;
                    mirror[n] = (i > c && j >= 0 && j < lw) ? j : -1;
                    ++n;
                }
            }
            *n_uniq = n;
            return DDSPP_OK;
        }
    }
    for (int i = 0; i < lw; ++i) {
        uniq[i] = i;
        mirror[i] = -1;
    }
    *n_uniq = lw;
    return DDSPP_OK;
}

// Even/odd tables of the full-window FIR design (ddspp_fir_from_magnitudes_eo, ddspp_frequency_filter_eo*):
//   CE[K/2, NJ], CO[K/2, NJ]: even / odd rows of the inverse real DFT at the NJ = (K-1)/2 + 1 distinct lags;
//   z[j] = E[j] + O[j], z[K-1-j] = E[j] - O[j];  tap i of the causal FIR = hann[i] * z[(i + K - 1) % (2K - 2)]:
//   tap_idx[NJ,4] (-1: unused), tap_we[NJ,4], tap_wo[NJ,4] = the (up to four) taps lag j feeds and their weights on E, O.
int ddspp_fir_eo_tables_host(int K, int window_size, float* CE, float* CO, int* tap_idx, float* tap_we, float* tap_wo) {
    int lw = 0, nj = 0;
    DDSPP_REQUIRE(CE && CO && tap_idx && tap_we && tap_wo, "fir_eo_tables_host: null buffer");
    DDSPP_REQUIRE(ddspp_fir_tables_shape(K, window_size, &lw, &nj) == DDSPP_OK && nj > 0,
                  "fir_eo_tables_host: K=%d window_size=%d is not an even/odd shape (see ddspp_fir_tables_shape)", K,
                  window_size);
    const int ir_size = 2 * (K - 1), half = K - 1, kh = K / 2;
    for (int r = 0; r < kh; ++r)
        for (int j = 0; j < nj; ++j) {
            const int ke = 2 * r, ko = 2 * r + 1;
            const double ce = ((ke == 0 || ke == K - 1) ? 1.0 : 2.0) * cos(2.0 * kPi * ke * (double)j / ir_size) / ir_size;
            const double co = ((ko == 0 || ko == K - 1) ? 1.0 : 2.0) * cos(2.0 * kPi * ko * (double)j / ir_size) / ir_size;
            CE[(size_t)r * nj + j] = (float)ce;
            CO[(size_t)r * nj + j] = (float)co;
        }
    std::vector<float> win(ir_size);
    hann_f32(ir_size, 1, win.data());
    std::vector<int> fill(nj, 0);
    for (int i = 0; i < nj * 4; ++i) {
        tap_idx[i] = -1;
        tap_we[i] = 0.0f;
        tap_wo[i] = 0.0f;
    }
    for (int i = 0; i < ir_size; ++i) {
        int jj = (i + half) % ir_size;
        if (jj > half) jj = ir_size - jj;
        int lane;
        float sign;
        if (jj <= half / 2) {
            lane = jj;
            sign = 1.0f;
        } else {
            lane = half - jj;
            sign = -1.0f;
        }
        const int s = fill[lane]++;
        DDSPP_REQUIRE(s < 4, "fir_eo_tables_host: internal error (more than four taps per lag)");
        tap_idx[lane * 4 + s] = i;
        tap_we[lane * 4 + s] = win[i];
        tap_wo[lane * 4 + s] = sign * win[i];
    }
    return DDSPP_OK;
}

}  // extern "C"
