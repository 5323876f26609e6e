// The polyphonic processor group as ONE call: processor_group(features, return_outputs_dict=True)
// (ddsp_piano/modules/piano_model.py:160) = the DAG of ddsp_piano/modules/polyphonic_dag.py:24-40 --
// MultiInharmonic.get_controls / get_signal, FilteredNoise, the MultiAdd chain, Reverb -- for P voices of B segments.
//
// ddspp_group_create builds, once per configuration, everything the kernels need besides the caller's tensors: the
// small tables (host builders of tables.cpp, uploaded here -- the only device memory the library allocates itself) and
// the rocFFT plan of the reverb.  ddspp_group_run enqueues the whole chain on the caller's stream inside a workspace
// the caller owns: get_controls over all rows -> compacted oscillator bank -> fused FilteredNoise with voice sums ->
// add chain -> reverb; with `outputs` also what the reference's outputs dictionary holds (the dry mix, the last
// voice's stems and conditioned controls, the last `add` node's first operand).  No host synchronisation, no
// allocation, nothing but kernel launches: the call can be captured in a HIP graph.  The Python layer's batched route
// (ddsp_piano_amd/polyphonic.py) enqueues the same kernels with the same arguments (tests/test_gpu_native_group.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <new>
#include <vector>

#include "ddspp_common.h"
#include "../../include/ddspp.h"

struct ddspp_group {
    ddspp_group_config c;
    int R, N, Lw, NJ, vpr;                 // rows, samples, FIR taps, even/odd table rows, voices per noise row
    bool fused_noise = true;
    // device tables
    float *wlin = nullptr, *whann = nullptr, *CE = nullptr, *CO = nullptr, *tap_we = nullptr, *tap_wo = nullptr;
    int* tap_idx = nullptr;
    ddspp_fftconv_plan* plan = nullptr;
    // workspace layout (byte offsets, 256-byte aligned)
    size_t o_amp, o_hd, o_aud, o_shl, o_addws, o_mix, o_alast, o_noise, o_zrows, o_zlast, o_zpack, o_dry, o_prev, o_fft, o_ir, total;
    size_t addws_bytes, fft_bytes;
    // the noise branch (and the impulse response's transform) does not depend on the additive branch until the mix: for
    // large batches it runs on a stream of the group's own, forked from and joined to the caller's with two events
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    uint64_t calls = 0;                    // counter of the library's own noise stream (one step per run, as the Python layer)
};

namespace {

size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename T>
int upload(T** dst, const std::vector<T>& host) {
    DDSPP_HIP_CHECK(hipMalloc((void**)dst, host.size() * sizeof(T)));
    DDSPP_HIP_CHECK(hipMemcpy(*dst, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return DDSPP_OK;
}

void release(ddspp_group* g) {
    if (!g) return;
    (void)hipFree(g->wlin); (void)hipFree(g->whann); (void)hipFree(g->CE); (void)hipFree(g->CO);
    (void)hipFree(g->tap_we); (void)hipFree(g->tap_wo); (void)hipFree(g->tap_idx);
    if (g->plan) ddspp_fftconv_plan_destroy(g->plan);
    if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
    if (g->ev_join) (void)hipEventDestroy(g->ev_join);
    if (g->side) (void)hipStreamDestroy(g->side);
    delete g;
}

}  // namespace

extern "C" {

int ddspp_group_create(const ddspp_group_config* cfg, ddspp_group** out) {
    DDSPP_REQUIRE(cfg && out, "group_create: null argument");
    const ddspp_group_config& c = *cfg;
    DDSPP_REQUIRE(c.n_segments > 0 && c.n_voices > 0 && c.n_frames >= 2 && c.n_harmonics > 0 && c.n_substrings > 0 &&
                      c.n_bands > 1 && c.upsampling > 0 && c.ir_length >= 0,
                  "group_create: bad dimensions");
    DDSPP_REQUIRE(c.upsampling % 8 == 0, "group_create: upsampling=%d must be a multiple of 8", c.upsampling);
    DDSPP_REQUIRE(c.n_voices * c.n_substrings <= 64, "group_create: n_voices * n_substrings = %d exceeds 64",
                  c.n_voices * c.n_substrings);
    DDSPP_REQUIRE(c.ir_batch == 0 || c.ir_batch == 1 || c.ir_batch == c.n_segments,
                  "group_create: ir_batch must be 1 or n_segments");
    ddspp_group* g = new (std::nothrow) ddspp_group();
    DDSPP_REQUIRE(g, "group_create: out of memory");
    g->c = c;
    const int B = c.n_segments, P = c.n_voices, T = c.n_frames, H = c.n_harmonics, S = c.n_substrings, K = c.n_bands,
              U = c.upsampling;
    g->R = B * P;
    g->N = T * U;
    const int N = g->N, R = g->R;
    int rc = ddspp_fir_tables_shape(K, c.window_size, &g->Lw, &g->NJ);
    if (rc == DDSPP_OK && g->NJ <= 0) {
        ddspp_set_error("group_create: n_bands=%d / window_size=%d has no even/odd FIR tables (K in {32, 64, 96, 128}, full window)",
                        K, c.window_size);
        rc = DDSPP_EINVAL;
    }
    if (rc != DDSPP_OK) {
        release(g);
        return rc;
    }
    // the fused FilteredNoise kernel (FIR design + time-varying FIR, voices summed in registers) when the shape fits it,
    // else the two-call form with per-voice rows (e.g. 16 kHz: more than 16 frames reach a window of 1024 samples)
    g->fused_noise = ddspp_frequency_filter_eo_supported(N, T, K, g->Lw, c.delay_compensation) != 0;
    g->vpr = 1;
    // (voices are summed in the kernel only when the rows alone give it workgroups enough -- a window is 30 frames, the chip
    // holds 768 workgroups: a single 3 s segment is 50 units of eight voices or 400 of one)
    const int forced = ddspp_option_literal("DDSPP_VOICE_SUMS", 0);                // (tests: voice sums at small sizes)
    if (g->fused_noise)
        for (int v : {8, 4, 2})
            if (P % v == 0 && (forced > 0 ? v <= forced : (long long)B * (P / v) * ((T + 29) / 30) >= 768)) {
                g->vpr = v;
                break;
            }
    // ---- tables -------------------------------------------------------------------------------------------------
    {
        std::vector<float> w(N), hann(2 * U);
        int walkable = 0;
        rc = ddspp_walk_weights_host(T, N, c.resize_rule, 0, N, w.data(), &walkable);
        if (rc == DDSPP_OK && !walkable) {
            ddspp_set_error("group_create: the bilinear source rows of T=%d -> N=%d do not follow the frame walk", T, N);
            rc = DDSPP_EINVAL;
        }
        if (rc == DDSPP_OK) rc = ddspp_hann_window_host(2 * U, hann.data());
        const int kh = K / 2, nj = g->NJ;
        std::vector<float> CE((size_t)kh * nj), CO((size_t)kh * nj), we((size_t)nj * 4), wo((size_t)nj * 4);
        std::vector<int> idx((size_t)nj * 4);
        if (rc == DDSPP_OK) rc = ddspp_fir_eo_tables_host(K, c.window_size, CE.data(), CO.data(), idx.data(), we.data(), wo.data());
        if (rc == DDSPP_OK) rc = upload(&g->wlin, w);
        if (rc == DDSPP_OK) rc = upload(&g->whann, hann);
        if (rc == DDSPP_OK) rc = upload(&g->CE, CE);
        if (rc == DDSPP_OK) rc = upload(&g->CO, CO);
        if (rc == DDSPP_OK) rc = upload(&g->tap_we, we);
        if (rc == DDSPP_OK) rc = upload(&g->tap_wo, wo);
        if (rc == DDSPP_OK) rc = upload(&g->tap_idx, idx);
    }
    if (rc == DDSPP_OK && c.ir_length > 0)
        rc = ddspp_fftconv_plan_create(B, c.ir_batch ? c.ir_batch : B, N, c.ir_length, &g->plan);
    // a side stream for the noise branch is opt-in (DDSPP_SIDE_STREAM=1): both branches want the same VALU issue slots
    if (rc == DDSPP_OK && ddspp_option_literal("DDSPP_SIDE_STREAM", 0) &&
        (long long)R * N >= (long long)ddspp_option_literal("DDSPP_SIDE_STREAM_MIN", 1 << 24) &&
        !ddspp_option_literal("DDSPP_NO_SIDE_STREAM", 0)) {
        hipError_t e = hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_join, hipEventDisableTiming);
        if (e != hipSuccess) {
            ddspp_set_error("group_create: side stream: %s", hipGetErrorString(e));
            rc = DDSPP_EHIP;
        }
    }
    if (rc != DDSPP_OK) {
        release(g);
        return rc;
    }
    // ---- workspace layout ---------------------------------------------------------------------------------------
    g->addws_bytes = ddspp_polyphonic_additive_workspace_bytes(B, P, T, S, H, U);
    g->fft_bytes = g->plan ? ddspp_fftconv_workspace_bytes(g->plan) : 0;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o = up256(o + bytes);
        return at;
    };
    g->o_amp = take((size_t)R * T * 4);
    g->o_hd = take((size_t)R * T * H * 4);
    g->o_aud = take((size_t)R * T * 4);
    g->o_shl = take((size_t)B * T * H * 4);
    g->o_addws = take(g->addws_bytes);
    g->o_mix = take((size_t)B * N * 4);
    g->o_alast = take((size_t)B * N * 4);
    g->o_noise = take(((size_t)R * N + 3) / 4 * 4 * 4);
    g->o_zrows = take((size_t)(R / g->vpr) * N * 4);
    g->o_zlast = take((size_t)B * N * 4);
    g->o_zpack = take(g->vpr == 1 ? (size_t)B * (P > 1 ? P - 1 : 1) * N * 4 : 0);
    g->o_dry = take((size_t)B * N * 4);
    g->o_prev = take((size_t)B * N * 4);
    g->o_fft = take(g->fft_bytes);
    g->o_ir = take(g->fused_noise ? 0 : (size_t)R * T * g->Lw * 4);
    g->total = o;
    *out = g;
    return DDSPP_OK;
}

void ddspp_group_destroy(ddspp_group* g) { release(g); }

// sizes of the two structs as this build of the library sees them: a binding checks its own declaration against them
size_t ddspp_group_config_bytes(void) { return sizeof(ddspp_group_config); }
size_t ddspp_group_outputs_bytes(void) { return sizeof(ddspp_group_outputs); }

size_t ddspp_group_workspace_bytes(const ddspp_group* g) { return g ? g->total : 0; }
int ddspp_group_n_samples(const ddspp_group* g) { return g ? g->N : -1; }

int ddspp_group_run(ddspp_group* g, const float* amplitudes, const float* harmonic_distribution, const float* inharm_coef,
                    const float* f0_hz, const float* magnitudes, const float* reverb_ir, const float* noise, float* audio,
                    const ddspp_group_outputs* outputs, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    DDSPP_REQUIRE(g && amplitudes && harmonic_distribution && inharm_coef && f0_hz && magnitudes && audio && workspace,
                  "group_run: null argument");
    DDSPP_REQUIRE(workspace_bytes >= g->total, "group_run: workspace too small (%zu < %zu)", workspace_bytes, g->total);
    DDSPP_REQUIRE((uintptr_t)workspace % 256 == 0, "group_run: workspace must be 256-byte aligned");
    DDSPP_REQUIRE(!g->plan == !reverb_ir, "group_run: reverb_ir %s", g->plan ? "is missing" : "given, but the group has no reverb");
    const ddspp_group_config& c = g->c;
    const int B = c.n_segments, P = c.n_voices, T = c.n_frames, H = c.n_harmonics, S = c.n_substrings, K = c.n_bands,
              U = c.upsampling, R = g->R, N = g->N, vm = c.voice_major ? 1 : 0, vpr = g->vpr;
    char* ws = (char*)workspace;
    float* amp_c = (float*)(ws + g->o_amp);
    float* hd_c = (float*)(ws + g->o_hd);
    int* aud = (int*)(ws + g->o_aud);
    float* mix = (float*)(ws + g->o_mix);
    // the dry mix: the caller's buffer when the dictionary wants it, else scratch (or, without a reverb, the audio itself)
    float* dry = (outputs && outputs->dry) ? outputs->dry : (g->plan ? (float*)(ws + g->o_dry) : audio);
    const bool want = outputs != nullptr;
    const int last_row0 = vm ? (P - 1) * B : P - 1;            // first row of the last voice; its rows are B apart x 1 (vm) or x P
    const size_t row_step = vm ? 1 : (size_t)P;                // in rows
    int rc;

    // ---- noise branch (filtered_noise_synth.py:27-42): draw, design + time-varying FIR, voices summed per row; and the
    //      reverb's impulse responses -> spectra (an input: ready long before the dry mix) -- on the side stream if any ------
    // Every argument has been validated above: nothing below fails on its inputs.  Should a launch fail all the same
    // after the fork, the side stream is joined before returning (leave()), so a stream capture stays well formed and
    // the caller's next use of the workspace is ordered behind the side stream's work.
    hipStream_t zs = stream;
    bool forked = false, joined = false;
    auto leave = [&](int code) {
        if (forked && !joined) {
            joined = true;
            if (hipEventRecord(g->ev_join, g->side) == hipSuccess) (void)hipStreamWaitEvent(stream, g->ev_join, 0);
        }
        return code;
    };
    // (DDSPP_HIP_CHECK returns on the spot; between the fork and the join a failing HIP call has to go through leave())
#define DDSPP_HIP_CHECK_LEAVE(expr)                                                                          \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) {                                                                              \
            ddspp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);      \
            return leave(DDSPP_EHIP);                                                                        \
        }                                                                                                    \
    } while (0)
    if (g->side) {
        DDSPP_HIP_CHECK(hipEventRecord(g->ev_fork, stream));
        forked = true;                      // from here on the side stream may hold work of this call
        DDSPP_HIP_CHECK_LEAVE(hipStreamWaitEvent(g->side, g->ev_fork, 0));
        zs = g->side;
    }
    if (g->plan && g->side) {
        rc = ddspp_fftconv_transform_ir(g->plan, reverb_ir, c.reverb_keep_dry_tap ? 0 : 1, ws + g->o_fft, g->fft_bytes, zs);
        if (rc != DDSPP_OK) return leave(rc);
    }
    const float* z = noise;
    // (round 6) no noise given and the windowed kernel takes the shape: it draws the numbers itself while staging them
    const bool draw = !z && g->fused_noise &&
                      ddspp_frequency_filter_eo_drawn_supported(N, T, K, g->Lw, c.delay_compensation);
    const unsigned long long draw_off = (unsigned long long)g->calls << 40;
    if (draw) ++g->calls;
    if (!z && !draw) {
        float* zbuf = (float*)(ws + g->o_noise);
        rc = ddspp_uniform_noise(zbuf, ((size_t)R * N + 3) / 4 * 4, c.noise_seed, g->calls << 40, zs);
        if (rc != DDSPP_OK) return leave(rc);
        ++g->calls;
        z = zbuf;
    }
    float* zrows = (float*)(ws + g->o_zrows);
    float* zlast = want ? (outputs->noise_last ? outputs->noise_last : (float*)(ws + g->o_zlast)) : nullptr;
    const bool split_in_kernel = want && vpr > 1;
    if (draw) {
        rc = ddspp_frequency_filter_eo_voices_drawn(c.noise_seed, draw_off, magnitudes, g->CE, g->CO, g->tap_idx, g->tap_we,
                                                    g->tap_wo, zrows, split_in_kernel ? zlast : nullptr, R, N, T, K, g->Lw,
                                                    g->NJ, c.delay_compensation, c.noise_scale_kind, c.noise_bias,
                                                    c.noise_exponent, c.noise_max_value, c.noise_threshold, c.noise_gain, P,
                                                    vpr, vm, zs);
    } else if (g->fused_noise) {
        rc = ddspp_frequency_filter_eo_voices(z, magnitudes, g->CE, g->CO, g->tap_idx, g->tap_we, g->tap_wo, zrows,
                                              split_in_kernel ? zlast : nullptr, R, N, T, K, g->Lw, g->NJ, c.delay_compensation,
                                              c.noise_scale_kind, c.noise_bias, c.noise_exponent, c.noise_max_value,
                                              c.noise_threshold, c.noise_gain, P, vpr, vm, zs);
    } else {
        float* ir = (float*)(ws + g->o_ir);
        rc = ddspp_fir_from_magnitudes_eo(magnitudes, g->CE, g->CO, g->tap_idx, g->tap_we, g->tap_wo, ir, (size_t)R * T, K, g->Lw,
                                          g->NJ, c.noise_scale_kind, c.noise_bias, c.noise_exponent, c.noise_max_value,
                                          c.noise_threshold, c.noise_gain, zs);
        if (rc == DDSPP_OK) rc = ddspp_time_varying_fir(z, ir, zrows, R, N, T, g->Lw, c.delay_compensation, zs);
    }
    if (rc != DDSPP_OK) return leave(rc);
    if (g->side) DDSPP_HIP_CHECK_LEAVE(hipEventRecord(g->ev_join, g->side));

    // ---- get_controls of the additive processor over all rows (inharm_synth.py:167-219, :254-270) -------------------
    float* shifts_last = want ? (outputs->harmonic_shifts_last ? outputs->harmonic_shifts_last : (float*)(ws + g->o_shl)) : nullptr;
    // (the only reader of hd_c is the compacted bank below, and the copy of the last voice's rows: the sparse form, round 6)
    rc = ddspp_inharmonic_controls_sparse(amplitudes, harmonic_distribution, inharm_coef, f0_hz, amp_c, hd_c,
                                         want ? shifts_last : nullptr, aud, R, T, H, S, P, vm, c.sample_rate,
                                         c.min_frequency, c.scale_kind, c.exponent, c.max_value, c.threshold, c.gain,
                                         c.normalize_after_nyquist_cut, c.normalize_below_nyquist, stream);
    if (rc != DDSPP_OK) return leave(rc);

    // ---- additive branch: the compacted oscillator bank (inharm_synth.py:272-293 -> :87-127 -> :49-84) ---------------
    float* add_last = want ? (outputs->additive_last ? outputs->additive_last : (float*)(ws + g->o_alast)) : nullptr;
    rc = ddspp_polyphonic_additive(f0_hz, amp_c, hd_c, nullptr, inharm_coef, aud, g->wlin, g->whann, nullptr, mix, add_last, B,
                                   P, T, S, H, U, c.sample_rate, 0, vm, ws + g->o_addws, g->addws_bytes, stream);
    if (rc != DDSPP_OK) return leave(rc);

    // ---- add chain (polyphonic_dag.py:28-37) -------------------------------------------------------------------------
    if (g->side) {
        DDSPP_HIP_CHECK_LEAVE(hipStreamWaitEvent(stream, g->ev_join, 0));
        joined = true;
    }
    if (!want) {
        // voice sums leave the noise kernel segment major ([B, P / vpr, N]); per-voice rows keep the controls' order
        rc = ddspp_mix_voices(mix, 1, zrows, P / vpr, dry, B, N, N, vpr > 1 ? 0 : vm, stream);
        if (rc != DDSPP_OK) return leave(rc);
    } else if (P == 1) {
        // one voice: it is the last one.  dry = noise + additive (the first `add` node, two operands)
        if (zlast != zrows) DDSPP_HIP_CHECK_LEAVE(hipMemcpyAsync(zlast, zrows, (size_t)B * N * 4, hipMemcpyDeviceToDevice, stream));
        rc = ddspp_mix_voices(add_last, 1, zlast, 1, dry, B, N, N, 0, stream);
        if (rc != DDSPP_OK) return leave(rc);
    } else {
        const float* zr = zrows;
        int pz = P / vpr, zvm = 0;
        if (!split_in_kernel) {            // per-voice noise rows (P has no even divisor): take the last voice out by copies
            if (vm) {                      // [P, B, N]: the first (P - 1) B rows are the other voices, the last B rows the last one
                DDSPP_HIP_CHECK_LEAVE(hipMemcpyAsync(zlast, zrows + (size_t)(P - 1) * B * N, (size_t)B * N * 4, hipMemcpyDeviceToDevice,
                                               stream));
                pz = P - 1;
                zvm = 1;
            } else {                       // [B, P, N]
                float* pack = (float*)(ws + g->o_zpack);
                DDSPP_HIP_CHECK_LEAVE(hipMemcpy2DAsync(pack, (size_t)(P - 1) * N * 4, zrows, (size_t)P * N * 4, (size_t)(P - 1) * N * 4, B,
                                                 hipMemcpyDeviceToDevice, stream));
                DDSPP_HIP_CHECK_LEAVE(hipMemcpy2DAsync(zlast, (size_t)N * 4, zrows + (size_t)(P - 1) * N, (size_t)P * N * 4, (size_t)N * 4, B,
                                                 hipMemcpyDeviceToDevice, stream));
                zr = pack;
                pz = P - 1;
            }
        }
        float* prev = outputs->prev ? outputs->prev : (float*)(ws + g->o_prev);
        rc = ddspp_mix_last_voice(mix, 1, zr, pz, zlast, add_last, prev, dry, B, N, zvm, stream);
        if (rc != DDSPP_OK) return leave(rc);
    }

    // ---- the last voice's conditioned controls, as the re-used processors hold them ----------------------------------
    if (want) {
        const size_t sp = row_step;
        if (outputs->amplitudes_last)
            DDSPP_HIP_CHECK_LEAVE(hipMemcpy2DAsync(outputs->amplitudes_last, (size_t)T * 4, amp_c + (size_t)last_row0 * T, sp * T * 4,
                                             (size_t)T * 4, B, hipMemcpyDeviceToDevice, stream));
        if (outputs->harmonic_distribution_last)
            DDSPP_HIP_CHECK_LEAVE(hipMemcpy2DAsync(outputs->harmonic_distribution_last, (size_t)T * H * 4, hd_c + (size_t)last_row0 * T * H,
                                             sp * T * H * 4, (size_t)T * H * 4, B, hipMemcpyDeviceToDevice, stream));
        if (outputs->magnitudes_last) {    // FilteredNoise.get_controls of the last voice: scale_fn(magnitudes + initial_bias)
            DDSPP_HIP_CHECK_LEAVE(hipMemcpy2DAsync(outputs->magnitudes_last, (size_t)T * K * 4, magnitudes + (size_t)last_row0 * T * K,
                                             sp * T * K * 4, (size_t)T * K * 4, B, hipMemcpyDeviceToDevice, stream));
            if (c.noise_scale_kind >= 0) {
                rc = ddspp_scale_bias(outputs->magnitudes_last, outputs->magnitudes_last, (size_t)B * T * K, c.noise_bias,
                                      c.noise_scale_kind, c.noise_exponent, c.noise_max_value, c.noise_threshold, c.noise_gain,
                                      stream);
                if (rc != DDSPP_OK) return leave(rc);
            }
        }
    }

    // ---- reverb (ddsp.effects.Reverb.get_signal: mask the dry tap, FFT convolution, + dry; or, reverb_keep_dry_tap,
    //      FeedbackDelayNetwork.get_signal, fdn_reverb.py:407-410: the FFT convolution alone) ---------------------------
    if (g->plan) {
        // [_mask_dry_ir +] fft_convolve(padding='same', delay_compensation=0) [+ dry]
        if (g->side)
            rc = ddspp_fftconv_execute_prepared(g->plan, dry, N, audio, N, 0, c.reverb_add_dry ? 1 : 0, ws + g->o_fft, g->fft_bytes,
                                                stream);
        else
            rc = ddspp_fftconv_execute(g->plan, dry, N, reverb_ir, audio, N, 0, c.reverb_keep_dry_tap ? 0 : 1, c.reverb_add_dry ? 1 : 0, ws + g->o_fft,
                                       g->fft_bytes, stream);
        if (rc != DDSPP_OK) return rc;
    } else if (dry != audio) {             // no reverb node: the group's signal is the dry mix
        DDSPP_HIP_CHECK(hipMemcpyAsync(audio, dry, (size_t)B * N * 4, hipMemcpyDeviceToDevice, stream));
    }
    return DDSPP_OK;
#undef DDSPP_HIP_CHECK_LEAVE
}

}  // extern "C"
