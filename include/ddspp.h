/* libddspp -- C-ABI of the MI355X (gfx950) DDSP-Piano synthesis hot path.
 *
 * The reference (lrenault/ddsp-piano @ v2) is pure Python over TensorFlow + ddsp==3.7.0 and has no
 * FFI of its own; each entry point below therefore names the reference
 * *function* (file:line under /root/reference, or the un-vendored ddsp 3.7.0 function reached
 * from that call site) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major float32 unless noted;
 *   - `R` ("rows") is batch x voices flattened, `T` control frames, `U` samples per frame,
 *     `N = T * U` audio samples, `H` harmonics, `S` sub-strings, `K` noise bands, `L` IR taps;
 *   - functions only enqueue work on `stream` and return 0, or a negative errno-style code after
 *     storing a message for ddspp_last_error(); nothing is thrown across the boundary;
 *   - the caller owns all buffers, including workspaces; the library owns only rocFFT plans
 *     behind ddspp_fftconv_plan and the table set behind ddspp_group.
 */
#ifndef DDSPP_H_
#define DDSPP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#define DDSPP_OK 0
#define DDSPP_EINVAL (-22)
#define DDSPP_ENOMEM (-12)
#define DDSPP_EHIP (-5)
#define DDSPP_EFFT (-6)

#define DDSPP_SCALE_NONE 0
#define DDSPP_SCALE_EXP_SIGMOID 1 /* ddsp.core.exp_sigmoid */
#define DDSPP_SCALE_EXP_TANH 2    /* ddsp_piano/modules/inharm_synth.py:13-17 */

int ddspp_version(void);
const char* ddspp_target_arch(void);
const char* ddspp_last_error(void);
/* Tuning options (launch geometry, A/B route switches): named like their DDSPP_* environment
 * variable.  The environment is consulted ONCE per option (first use) and cached -- no getenv on the call path;
 * ddspp_set_option overrides a value, ddspp_reload_options forgets the cache (next use reads the environment again).
 * One option is not tuning but a RECALLED DETAIL of ddsp.core.angular_cumsum (`angular_offsets`): the
 * running sum of chunk end phases is added to a chunk wrapped, `tf.cumsum(offsets, axis=1) % (2 pi)` (the default, 0), or as it
 * is (DDSPP_ANGULAR_OFFSETS_PLAIN = 1).  Every entry point that runs the angular cumsum reads it at launch
 * (ddspp_cos_oscillator_bank, ddspp_harmonic_synthesis, ddspp_surrogate_harmonic_synthesis, ddspp_polyphonic_additive,
 * ddspp_polyphonic_stems, ddspp_polyphonic_surrogate_additive, ddspp_group_run); a caller that resets the options sets it again. */
int ddspp_option(const char* name, int default_value);      /* the value in effect */
int ddspp_set_option(const char* name, int value);
void ddspp_reload_options(void);

/* ---- host builders of the kernels' small tables (csrc/tables.cpp) --------------------------------------------
 * Everything below that takes a table (lo/hi/w, wlin, whann, window, M, CE/CO/tap_*) gets it from here: a caller that
 * binds libddspp.so without the Python layer needs nothing else.  Outputs are HOST buffers, the caller uploads them.
 * The `rule` arguments select between recollections of the un-vendored ddsp / TF details;
 * 0 is the default everywhere. */

/* tf.signal.hann_window(n) (periodic), float32 arithmetic: window[n].  `whann` of ddspp_harmonic_synthesis /
 * ddspp_polyphonic_additive and `window` of ddspp_resample_window are ddspp_hann_window_host(2 * U, ...). */
int ddspp_hann_window_host(int n, float* window);
/* Source rows and weights of ddsp.core.resample(method='linear') = tf.compat.v1.image.resize(BILINEAR,
 * align_corners=False): lo[N], hi[N] (int32), w[N] (float32).  rule 0: TF1 legacy kernel, pos = n * T/N;
 * rule 1: half-pixel centres.  *aligned == 1: N = T * U and lo[n] == n / U everywhere (rule 0, every shipped sample /
 * frame rate pair, up to 131 072 frames at hop 96); `wlin` of ddspp_harmonic_synthesis / ddspp_polyphonic_additive is
 * the table of ddspp_walk_weights_host (== w while aligned). */
int ddspp_resample_tables_host(int T, int N, int rule, int* lo, int* hi, float* w, int* aligned);
/* w[n] for samples first_sample .. first_sample + n - 1 of a signal with T frames per N samples: `wlin` of a streamed
 * piece (ddspp_polyphonic_additive / ddspp_oscillator_phase_state with phase_state_in).  The resize kernel multiplies
 * float32(sample index) by the float32 scale at the ABSOLUTE index; the fractional part rounds differently at
 * different magnitudes, so a piece takes the weights the one-call render has at its positions. */
int ddspp_linear_weights_host(int T, int N, int rule, long long first_sample, int n, float* w);
/* `wlin` as ddspp_harmonic_synthesis / ddspp_polyphonic_additive / ddspp_oscillator_phase_state take it: w of
 * ddspp_linear_weights_host, with the mark 1.0 on the samples whose source rows are (t + 1, t + 1) instead of (t, t + 1)
 * -- far into a long file float32(n) * float32(T / N) rounds up to the next whole frame for the last sample(s) of a
 * frame (from frame 131 073 at hop 96: synthesize_midi_file.py:41-73 on a piece of more than 8.7 minutes) and the
 * reference's resize then takes x[t + 1] itself; the kernels substitute it exactly.  *walkable (may be NULL) = 0 when
 * the frame walk cannot reproduce the table (N % T != 0, half-pixel rule, hours-long signals): use the three-operator
 * route (ddspp_resample_linear + ddspp_cos_oscillator_bank) then. */
int ddspp_walk_weights_host(int T, int N, int rule, long long first_sample, int n, float* w, int* walkable);
/* FIR length Lw of ddsp.core.frequency_impulse_response(magnitudes[.., K], window_size) and the row count NJ of the
 * even/odd tables (0: the shape has none -- use ddspp_fir_matrix_host + ddspp_fir_from_magnitudes). */
int ddspp_fir_tables_shape(int K, int window_size, int* Lw, int* NJ);
/* M[K,Lw] of ddspp_fir_from_magnitudes; uniq[Lw], mirror[Lw], *n_uniq (all three may be NULL): its symmetry lists.
 * crop_rule 0: apply_window_to_impulse_response's crop indices as recalled from ddsp 3.7.0; 1: centred crop. */
int ddspp_fir_matrix_host(int K, int window_size, int crop_rule, float* M, int* uniq, int* mirror, int* n_uniq);
/* CE[K/2,NJ], CO[K/2,NJ], tap_idx[NJ,4], tap_we[NJ,4], tap_wo[NJ,4] of ddspp_fir_from_magnitudes_eo and
 * ddspp_frequency_filter_eo* (K in {32, 64, 96, 128}, full-length window). */
int ddspp_fir_eo_tables_host(int K, int window_size, float* CE, float* CO, int* tap_idx, float* tap_we, float* tap_wo);

/* ---- frame -> sample control upsamplers ------------------------------------------------------ */

/* ddsp.core.resample(x, N, method='linear') -- call site inharm_synth.py:117.
 * x[R,T,C] -> y[R,N,C];  y[n] = x[lo[n]] + (x[hi[n]] - x[lo[n]]) * w[n]  with the legacy-bilinear
 * (align_corners=False, no half-pixel) tables lo/hi (int32[N]) and w (float32[N]). */
int ddspp_resample_linear(const float* x, const int* lo, const int* hi, const float* w, float* y, int R,
                          int T, int C, int N, hipStream_t stream);

/* ddsp.core.resample(x, N, method='window') = ddsp.core.upsample_with_windows(add_endpoint=True)
 * -- call site inharm_synth.py:118-119.  window = tf.signal.hann_window(2U) as float32[2U]. */
int ddspp_resample_window(const float* x, const float* window, float* y, int R, int T, int C, int U,
                          hipStream_t stream);

/* surrogate_harmonic_synthesis' decay envelope (ddsp_piano/modules/surrogate_synth.py:76-95), in place:
 * amplitude_envelopes[R,T*U,C] *= |decays[R,T,C]| ** (decay_time[R,T] * U + n % U). */
int ddspp_decay_envelope(float* amplitude_envelopes, const float* decays, const float* decay_time, int R, int T,
                         int C, int U, hipStream_t stream);

/* ---- oscillator bank ------------------------------------------------------------------------- */

size_t ddspp_osc_workspace_bytes(int R, int N, int V);

/* cos_oscillator_bank(frequency_envelopes, amplitude_envelopes, sample_rate, sum_sinusoids,
 * use_angular_cumsum) -- inharm_synth.py:49-84 (+ ddsp.core.remove_above_nyquist / angular_cumsum).
 * envelopes [R,N,H] -> audio [R,N] (sum_sinusoids) or [R,N,H].  spans = 0 lets the library pick how
 * many runs of 1000-sample chunks each row is cut into (1 = every envelope read exactly once). */
int ddspp_cos_oscillator_bank(const float* frequency_envelopes, const float* amplitude_envelopes,
                              float* audio, int R, int N, int H, float sample_rate, int sum_sinusoids,
                              int use_angular_cumsum, int spans, void* workspace, size_t workspace_bytes,
                              hipStream_t stream);

/* harmonic_synthesis(...) summed over sub-strings = MultiInharmonic.get_signal
 * -- inharm_synth.py:87-127, :221-244, :272-293.
 * f0_hz[R,T,S], amplitudes[R,T], harmonic_distribution[R,T,H], harmonic_shifts[R,T,H] (or NULL)
 * -> audio[R, T*U].  wlin = float32[T*U] linear weights, whann = float32[2U] Hann window. */
int ddspp_harmonic_synthesis(const float* f0_hz, const float* amplitudes,
                             const float* harmonic_distribution, const float* harmonic_shifts,
                             const float* wlin, const float* whann, float* audio, int R, int T, int S,
                             int H, int U, float sample_rate, int use_angular_cumsum, int spans,
                             void* workspace, size_t workspace_bytes, hipStream_t stream);

/* surrogate_harmonic_synthesis(...) = SurrogateAdditive.get_signal -- ddsp_piano/modules/surrogate_synth.py:11-104,
 * :203-214 (configs/surrogate.gin): harmonic_synthesis with every partial's amplitude envelope multiplied by
 * |decays[t,k]| ** (decay_time[t] * U + n % U), t = n / U (:76-95), straight from the frame-rate controls (the [R,N,H]
 * envelopes are never formed; ddspp_decay_envelope is the materialised form of the same term).
 * f0_hz[R,T,1], amplitudes[R,T], harmonic_distribution[R,T,H], harmonic_shifts[R,T,H] (or NULL), decays[R,T,H],
 * decay_time[R,T] -> audio[R, T*U].  Tables, workspace (ddspp_osc_workspace_bytes(R, T*U, H)) and spans as
 * ddspp_harmonic_synthesis. */
int ddspp_surrogate_harmonic_synthesis(const float* f0_hz, const float* amplitudes, const float* harmonic_distribution,
                                       const float* harmonic_shifts, const float* decays, const float* decay_time,
                                       const float* wlin, const float* whann, float* audio, int R, int T, int H, int U,
                                       float sample_rate, int use_angular_cumsum, int spans, void* workspace,
                                       size_t workspace_bytes, hipStream_t stream);

/* The additive branch of the whole polyphonic group: audio[B, T*U] = sum over the P voices of a segment
 * of MultiInharmonic.get_signal (the `additive/signal` terms of polyphonic_dag.py:28-37); rows of the
 * controls are [B * P] segment major (voice_major = 0), or [P * B] voice major (voice_major = 1): the layout the
 * reference's Parallelizer.unparallelize leaves the merged controls in (sub_modules.py:573-592), so the per-voice
 * keys `<name>_<i>` can be handed over without a copy.  Only oscillators with a non-zero amplitude somewhere in a
 * span are given a lane, so the work follows the number of partials below Nyquist instead of P * H.
 * harmonic_shifts NULL and inharm_coef[R,T] given (the raw get_controls input): the kernels form
 * harmonic_shifts = sqrt(k^2 max(inharm_coef, 0) + 1) - 1 (get_inharmonic_freq, inharm_synth.py:37-44) per lane and
 * frame -- bit for bit what ddspp_inharmonic_controls writes -- and the [R,T,H] tensor need not exist.
 * audible[R,T] (may be NULL): ddspp_inharmonic_controls' per-frame count of leading non-silent harmonics.  When given, a
 * harmonic at or above its frame's count is taken as silent from the count (its amplitude product is zero by definition)
 * and harmonic_distribution is not trusted there (ddspp_inharmonic_controls_sparse leaves it unwritten).
 * phase_state_in[R, S*H] (may be NULL): streaming -- these controls continue a signal; every oscillator starts from the
 * float32 running sum of chunk end phases the previous call left (ddspp_oscillator_phase_state).  The call must start
 * on a 1000-sample chunk boundary of the whole signal (a multiple of lcm(U, 1000) / U frames), and wlin then holds the
 * weights of the piece's absolute sample positions (ddspp_linear_weights_host).
 * audio_last[B, T*U] (may be NULL): when given, the LAST voice's stem goes there and `audio` holds the sum of voices
 * 0 .. P-2 -- the reference's DAG re-uses one additive processor for all voices, so its outputs dictionary keeps the
 * last voice's signal next to the mix (polyphonic_dag.py:28-37, piano_model.py:160-164). */
size_t ddspp_polyphonic_additive_workspace_bytes(int B, int P, int T, int S, int H, int U);
int ddspp_polyphonic_additive(const float* f0_hz, const float* amplitudes, const float* harmonic_distribution,
                              const float* harmonic_shifts, const float* inharm_coef, const int* audible,
                              const float* wlin, const float* whann, const float* phase_state_in, float* audio,
                              float* audio_last, int B, int P, int T, int S, int H, int U, float sample_rate, int spans,
                              int voice_major, void* workspace, size_t workspace_bytes, hipStream_t stream);

/* Every voice's stem of the same call: stems[B * P, T*U] = MultiInharmonic.get_signal of every voice, rows in the order of
 * the controls -- what synthesize_from_csv.py:99-120 obtains by calling the additive processor once per voice
 * (--decompose), and what the outputs dictionary of default_model.py:56-74's node list holds (every `sub_add_i` names a
 * voice's signal).  The same compacted bank with the harmonic sum stopped at voice boundaries: lanes only for audible
 * oscillators, the voices of a segment packed back to back in whole blocks of 32.  Arguments as
 * ddspp_polyphonic_additive (no streaming state); workspace: ddspp_polyphonic_stems_workspace_bytes.
 * Memory: the workspace holds one partial row of T*U floats per 32-entry block of the packed list, sized for the worst
 * case (every harmonic of every voice audible): B * P * S * ceil(H / 32) rows = ceil(H / 32) times the stems themselves
 * (H = 128: four times; config 3: 1.2 GB beside 0.3 GB of stems; config 5 at batch 256: 18.9 GB beside 4.7 GB) -- the caller
 * owns it and may release it after the call; ask ddspp_polyphonic_stems_workspace_bytes before choosing this entry point
 * for hours-long single segments. */
size_t ddspp_polyphonic_stems_workspace_bytes(int B, int P, int T, int S, int H, int U);
int ddspp_polyphonic_stems(const float* f0_hz, const float* amplitudes, const float* harmonic_distribution,
                           const float* harmonic_shifts, const float* inharm_coef, const int* audible, const float* wlin,
                           const float* whann, float* stems, int B, int P, int T, int S, int H, int U, float sample_rate,
                           int spans, int voice_major, void* workspace, size_t workspace_bytes, hipStream_t stream);

/* The same for SurrogateAdditive voices (surrogate_synth.py:11-104, configs/surrogate.gin through polyphonic_dag.py): every
 * partial's amplitude multiplied by |decays[t,k]| ** (decay_time[t] * U + n % U) inside the compacted bank.  decays[B*P,T,H]
 * as ddspp_surrogate_decays leaves them, decay_time[B*P,T]; one sub-string; workspace as ddspp_polyphonic_additive (S = 1). */
int ddspp_polyphonic_surrogate_additive(const float* f0_hz, const float* amplitudes, const float* harmonic_distribution,
                                        const float* harmonic_shifts, const float* inharm_coef, const int* audible,
                                        const float* decays, const float* decay_time, const float* wlin, const float* whann,
                                        float* audio, float* audio_last, int B, int P, int T, int H, int U, float sample_rate,
                                        int spans, int voice_major, void* workspace, size_t workspace_bytes,
                                        hipStream_t stream);

/* Streaming state of the oscillator banks (synthesize_midi_file.py:41-73 renders minutes of audio; this lets a caller do
 * it piecewise, or shard one file's TIME over several GPUs).  ddsp.core.angular_cumsum restarts the phase every 1000
 * samples and adds the float32 running sum of the chunks' end phases, so the only thing a later piece of the same signal
 * needs is that sum per (row, oscillator): phase_state_out[R, S*H] = phase_state_in (NULL: 0) + the end phases of the
 * first n_chunks chunks of these controls, added sequentially in float32 -- bit for bit what one long call holds there.
 * Controls as ddspp_polyphonic_additive (harmonic_shifts, or inharm_coef, or neither + harmonic_distribution as a dummy
 * [R,T,H] buffer); T * U >= n_chunks * 1000. */
size_t ddspp_oscillator_phase_state_workspace_bytes(int R, int S, int H, int n_chunks);
int ddspp_oscillator_phase_state(const float* f0_hz, const float* harmonic_shifts, const float* inharm_coef,
                                 const float* harmonic_distribution, const int* audible, const float* wlin,
                                 const float* phase_state_in, float* phase_state_out, int R, int T, int S, int H, int U,
                                 float sample_rate, int n_chunks, void* workspace, size_t workspace_bytes,
                                 hipStream_t stream);

/* ---- get_controls ---------------------------------------------------------------------------- */

/* InHarmonic.get_controls / MultiInharmonic.get_controls -- inharm_synth.py:167-219, :254-270
 * (+ get_inharmonic_freq :20-46).  amplitudes[R,T], harmonic_distribution[R,T,H], inharm_coef[R,T],
 * f0_hz[R,T,S] -> amplitudes_out[R,T] (already divided by S), harmonic_distribution_out[R,T,H],
 * harmonic_shifts_out[R,T,H] (may be NULL when the consumer forms the shifts from inharm_coef itself, see
 * ddspp_polyphonic_additive); audible_out[R,T] (int32, may be NULL): bits 0-15 = 1 + index of the last harmonic
 * with amplitudes_out * harmonic_distribution_out != 0 in the frame, bit 16 = some f0_hz sub-string or the clamped
 * inharm_coef differs from the previous frame's (the harmonic frequencies may have moved).
 * ddspp_polyphonic_additive accepts it as `audible`, so that it need not scan the [R,T,H] tensors again.
 * normalize_after_nyquist_cut: 1 = the distribution is normalised after the Nyquist cut (:210-214), 0 = before it
 * (:194-198), 2 = never -- SurrogateAdditive.get_controls with normalize_harm_distribution=False
 * (surrogate_synth.py:152-187 is this function with that flag and the decays of ddspp_surrogate_decays). */
int ddspp_inharmonic_controls(const float* amplitudes, const float* harmonic_distribution,
                              const float* inharm_coef, const float* f0_hz, float* amplitudes_out,
                              float* harmonic_distribution_out, float* harmonic_shifts_out, int* audible_out,
                              int R, int T, int H, int S, float sample_rate, float min_frequency, int scale_kind,
                              float exponent, float max_value, float threshold, float gain,
                              int normalize_after_nyquist_cut, int normalize_below_nyquist,
                              hipStream_t stream);
/* SurrogateAdditive.get_controls' decay factors -- ddsp_piano/modules/surrogate_synth.py:163-171:
 * decays_out[R,T,H] = where(inharmonic_freq >= sample_rate / 2, 1, clip(decays, 1e-5, 1)), the inharmonic frequencies formed
 * from f0_hz[R,T] and inharm_coef[R,T] as get_inharmonic_freq does. */
int ddspp_surrogate_decays(const float* decays, const float* inharm_coef, const float* f0_hz, float* decays_out, int R, int T,
                           int H, float sample_rate, hipStream_t stream);

/* The same over the R = n_segments * n_voices rows of a polyphonic group (segment major, or voice major as
 * ddspp_polyphonic_additive), writing harmonic_shifts only for every segment's LAST voice (shifts_last_out
 * [R / n_voices, T, H], may be NULL): ddspp_polyphonic_additive forms the shifts of all voices itself from inharm_coef,
 * the outputs dictionary of the reference's DAG keeps the last voice's controls (polyphonic_dag.py:28-37). */
int ddspp_inharmonic_controls_group(const float* amplitudes, const float* harmonic_distribution,
                                    const float* inharm_coef, const float* f0_hz, float* amplitudes_out,
                                    float* harmonic_distribution_out, float* shifts_last_out, int* audible_out,
                                    int R, int T, int H, int S, int n_voices, int voice_major, float sample_rate,
                                    float min_frequency, int scale_kind, float exponent, float max_value,
                                    float threshold, float gain, int normalize_after_nyquist_cut,
                                    int normalize_below_nyquist, hipStream_t stream);

/* ddspp_inharmonic_controls_group for a caller whose ONLY reader of harmonic_distribution_out is the compacted oscillator
 * bank (ddspp_polyphonic_additive / ddspp_polyphonic_stems / ddspp_polyphonic_surrogate_additive called with `audible` =
 * audible_out, which is required here).  A frame's row is written only below its audible count, in whole groups of 16
 * harmonics: every value left out is one whose product with amplitudes_out is exactly zero (that is what the count
 * means), and the bank takes those harmonics as silent from the count itself.  Rows of every segment's last voice are
 * written whole when shifts_last_out is given (the outputs dictionary keeps that voice's controls).  What is not written
 * keeps whatever the buffer held: harmonic_distribution_out is then NOT a valid tensor for any other reader.  At a
 * piano's note mix two thirds of the [R,T,H] write disappear (InHarmonic.get_controls, inharm_synth.py:200-214, zeroes
 * them).  Shapes or flags the lean kernel does not take are written whole, which is always valid. */
int ddspp_inharmonic_controls_sparse(const float* amplitudes, const float* harmonic_distribution,
                                     const float* inharm_coef, const float* f0_hz, float* amplitudes_out,
                                     float* harmonic_distribution_out, float* shifts_last_out, int* audible_out,
                                     int R, int T, int H, int S, int n_voices, int voice_major, float sample_rate,
                                     float min_frequency, int scale_kind, float exponent, float max_value,
                                     float threshold, float gain, int normalize_after_nyquist_cut,
                                     int normalize_below_nyquist, hipStream_t stream);

/* ddsp.synths.FilteredNoise.get_controls: y = scale_fn(x + initial_bias), elementwise. */
int ddspp_scale_bias(const float* x, float* y, size_t n, float bias, int scale_kind, float exponent,
                     float max_value, float threshold, float gain, hipStream_t stream);

/* ---- mixers ---------------------------------------------------------------------------------- */

/* MultiAdd.get_signal (inharm_synth.py:308-309) / ddsp.processors.Add: out = ((s0 + s1) + s2) ...
 * srcs = DEVICE array of nsrc device pointers. */
int ddspp_add_signals(const float* const* srcs, int nsrc, float* out, size_t n, hipStream_t stream);

/* the whole `add` chain of polyphonic_dag.py:28-37: additive/noise [B,P,N] (voice_major = 0) or [P,B,N]
 * (voice_major = 1) -> out rows of out_stride floats (noise may be NULL).  out_prev (may be NULL) [B,N]: the running
 * mix before the last voice, i.e. the first operand of the last `add` node. */
int ddspp_polyphonic_mix(const float* additive, const float* noise, float* out, float* out_prev, int B, int P, int N,
                         int out_stride, int voice_major, hipStream_t stream);

/* the whole add chain of default_model.py:56-74 with every node's signal (that node list NAMES its Add nodes, so the
 * reference's outputs dictionary holds them all): additive / noise [B,P,N] (voice_major = 0) or [P,B,N] -> sub[B,P,N],
 * sub[b, v] = noise[b, v] + additive[b, v] (`sub_add_v`; v = 0: `add_0`), and run[B,P,N], run[b, v] = run[b, v-1] + sub[b, v]
 * (`add_v`), both segment major.  One pass over the stems. */
int ddspp_add_chain_paired(const float* additive, const float* noise, float* sub, float* run, int B, int P, int N,
                           int voice_major, hipStream_t stream);

/* out[b] = sum of the PA rows a[b, :] + the PZ rows z[b, :] ([B,PA,N], [B,PZ,N], or [PA,B,N], [PZ,B,N] when
 * voice_major = 1 -> rows of out_stride floats): the add chain when the additive operand is already the
 * per-segment mix of ddspp_polyphonic_additive. */
int ddspp_mix_voices(const float* a, int PA, const float* z, int PZ, float* out, int B, int N, int out_stride,
                     int voice_major, hipStream_t stream);

/* The end of the add chain with the last voice kept apart -- what the reference's outputs dictionary holds for the
 * re-used processors (polyphonic_dag.py:34-37, piano_model.py:160-164): prev[b] = sum of the PA rows a[b, :] + the PZ
 * rows z[b, :] (voices 0 .. P-2), dry[b] = (prev[b] + noise_last[b]) + additive_last[b].  prev, dry, noise_last,
 * additive_last: [B, N]. */
int ddspp_mix_last_voice(const float* a, int PA, const float* z, int PZ, const float* noise_last,
                         const float* additive_last, float* prev, float* dry, int B, int N, int voice_major,
                         hipStream_t stream);

/* The same end of the chain for the node list of ddsp_piano/default_model.py:44-80 (explicit ddsp.processors.Add nodes):
 * sub[b] = noise_last[b] + additive_last[b] (`sub_add_{P-1}`, :68-70), dry[b] = prev[b] + sub[b] (`add_{P-1}`, :72-74),
 * prev = `add_{P-2}`.  All [B, N]. */
int ddspp_mix_last_voice_paired(const float* a, int PA, const float* z, int PZ, const float* noise_last,
                                const float* additive_last, float* prev, float* sub, float* dry, int B, int N,
                                int voice_major, hipStream_t stream);

/* ---- measurement aid --------------------------------------------------------------------------- */

/* A pure read of x[0 .. n_floats) by n_waves concurrent wavefront streams (1 KB per instruction, sixteen in flight: the
 * access pattern of ddspp_cos_oscillator_bank on materialised envelopes, nothing else) -- the HBM rate the graded kernel
 * could reach at most on that buffer in this run (bench.py: roofline.measured_peak).  sink: one float of device memory
 * (never written for finite data); *bytes_read (host, may be NULL): the bytes the launch reads. */
int ddspp_hbm_read_probe(const float* x, size_t n_floats, int n_waves, float* sink, size_t* bytes_read, hipStream_t stream);
/* The write-side counterpart: x[0 .. n_floats) filled by n_waves concurrent wavefront streams with 16-byte stores (1 KB per
 * instruction), non-temporal (1) or plain (0) -- the rate the stand-alone upsamplers (ddspp_resample_linear / _window, pure
 * write streams) could reach at most in this run (bench.py: extras.three_operator_chain.write_ceiling). */
int ddspp_hbm_write_probe(float* x, size_t n_floats, int n_waves, int nontemporal, size_t* bytes_written, hipStream_t stream);
/* Measurement aid for the VALU-bound kernels of the step (bench.py: roofline_step.measured_ceiling): a pure stream of
 * independent wave64 multiply-adds, `iters` x 192 per lane, on `waves_per_simd` wavefronts per SIMD of the whole chip;
 * *wave_fmas_per_simd = instructions each SIMD executed.  Timed by the caller (HIP events, ~20 ms so that the power
 * management settles): what the chip sustains, as opposed to one instruction per 2 cycles at the nominal clock. */
int ddspp_fma_probe(float* sink, int waves_per_simd, int iters, double* wave_fmas_per_simd, hipStream_t stream);

/* ---- FilteredNoise --------------------------------------------------------------------------- */

/* ddsp.core.frequency_impulse_response(magnitudes[frames,K], window_size) as magnitudes @ M with the
 * host-built float32 matrix M[K,Lw] -> ir[frames,Lw].  uniq/mirror (device int32[n_uniq], or NULL):
 * the taps to evaluate and the tap each result is mirrored to (-1: none) -- the FIRs are symmetric. */
int ddspp_fir_from_magnitudes(const float* magnitudes, const float* M, const int* uniq, const int* mirror,
                              int n_uniq, float* ir, size_t frames, int K, int Lw, hipStream_t stream);

/* The same operator for the full-window case (2 (K - 1) <= window_size), through the even/odd split of
 * the inverse real DFT: CE/CO[K/2, NJ] cosine tables, tap_idx/tap_we/tap_wo[NJ, 4]: the up-to-four taps
 * lane j produces and their weights, ir[tap] = we * E[j] + wo * O[j].
 * scale_kind 0 / 1 / 2 (none / exp_sigmoid / exp_tanh, as ddspp_scale_bias): `magnitudes` are the raw network
 * outputs and FilteredNoise.get_controls' scale_fn(magnitudes + bias) is applied on the way in;
 * scale_kind -1: magnitudes are used as given (bias and the scale parameters are ignored). */
int ddspp_fir_from_magnitudes_eo(const float* magnitudes, const float* CE, const float* CO, const int* tap_idx,
                                 const float* tap_we, const float* tap_wo, float* ir, size_t frames, int K,
                                 int Lw, int NJ, int scale_kind, float bias, float exponent, float max_value,
                                 float threshold, float gain, hipStream_t stream);

/* ddsp.core.fft_convolve(audio[R,N], impulse_response[R,T,Lw], padding='same', delay_compensation)
 * in the framed case (frame = hop = N / T); reached from filtered_noise_synth.py:41-42 through
 * ddsp.core.frequency_filter.  delay_compensation >= 0: that many leading samples are cropped;
 * DDSPP_DELAY_AUTO (-1): crop_and_compensate_delay's automatic start as recalled from ddsp 3.7.0,
 * (Lw - 1) // 2 - 1; DDSPP_DELAY_AUTO_HALF (-2): the alternative recollection Lw // 2 (DESIGN.md section 2 --
 * ddsp is not on disk, so the rule is a documented switch; every entry point that takes a delay accepts both). */
int ddspp_time_varying_fir(const float* audio, const float* impulse_response, float* out, int R, int N,
                           int T, int Lw, int delay_compensation, hipStream_t stream);

/* ddsp.core.frequency_filter(audio, magnitudes, window_size) of DynamicSizeFilteredNoise.get_signal
 * (filtered_noise_synth.py:27-42) in ONE kernel: the FIR design of ddspp_fir_from_magnitudes_eo (matrix cores) and
 * the time-varying FIR of ddspp_time_varying_fir, with the impulse responses kept in LDS -- the [R,T,Lw] tensor
 * is never written.  Same tables and scale arguments as ddspp_fir_from_magnitudes_eo; results equal the two-call
 * form bit for bit.  ddspp_frequency_filter_eo_supported tells whether a shape fits (at most 16 frames reach a
 * window of 1024 outputs, K in {32, 64, 96}, full window); otherwise use the two calls. */
int ddspp_frequency_filter_eo_supported(int N, int T, int K, int Lw, int delay_compensation);
int ddspp_frequency_filter_eo(const float* audio, const float* magnitudes, const float* CE, const float* CO,
                              const int* tap_idx, const float* tap_we, const float* tap_wo, float* out, int R,
                              int N, int T, int K, int Lw, int NJ, int delay_compensation, int scale_kind,
                              float bias, float exponent, float max_value, float threshold, float gain,
                              hipStream_t stream);
/* The same for the R = n_segments * n_voices rows of a polyphonic group (segment major, or voice major as
 * ddspp_polyphonic_additive), with the filtered noise of `voices_per_row` consecutive voices of a segment summed in
 * registers into ONE output row: out[R / voices_per_row, N] -- the noise half of the `add` chain of
 * polyphonic_dag.py:28-37 without a [R, N] round trip; ddspp_mix_voices adds the rows to the additive mix.
 * out_last (may be NULL; needs voices_per_row > 1) [R / n_voices, N]: every segment's LAST voice goes there instead
 * of into its row's sum (the reference's outputs dictionary keeps the re-used noise processor's last signal);
 * ddspp_mix_last_voice then finishes the chain. */
int ddspp_frequency_filter_eo_voices(const float* audio, const float* magnitudes, const float* CE, const float* CO,
                                     const int* tap_idx, const float* tap_we, const float* tap_wo, float* out,
                                     float* out_last,
                                     int R, int N, int T, int K, int Lw, int NJ, int delay_compensation,
                                     int scale_kind, float bias, float exponent, float max_value, float threshold,
                                     float gain, int n_voices, int voices_per_row, int voice_major,
                                     hipStream_t stream);
/* DynamicSizeFilteredNoise.get_signal with its own draw (filtered_noise_synth.py:39-42: tf.random.uniform, then
 * frequency_filter) in ONE kernel: the windowed kernel draws the U(-1, 1) numbers it filters while staging them --
 * exactly the ones ddspp_uniform_noise(buf, R * N, seed, offset) would have written into a [R, N] tensor (same
 * Philox4x32-10 counters, same bits), which then never exists (config 3: 0.29 GB written and read back per step).
 * Everything else as ddspp_frequency_filter_eo_voices (voices_per_row = 1, n_voices = 1: plain rows).
 * ddspp_frequency_filter_eo_drawn_supported: the shapes of the windowed kernel (every shipped hop / band count). */
int ddspp_frequency_filter_eo_drawn_supported(int N, int T, int K, int Lw, int delay_compensation);
int ddspp_frequency_filter_eo_voices_drawn(uint64_t seed, uint64_t offset, const float* magnitudes, const float* CE,
                                           const float* CO, const int* tap_idx, const float* tap_we, const float* tap_wo,
                                           float* out, float* out_last, int R, int N, int T, int K, int Lw, int NJ,
                                           int delay_compensation, int scale_kind, float bias, float exponent,
                                           float max_value, float threshold, float gain, int n_voices, int voices_per_row,
                                           int voice_major, hipStream_t stream);

/* NoiseBandNetSynth.get_signal -- filtered_noise_synth.py:213-262: audio[r, n] = sum_k noise_bands[(n mod noise_len
 * - shift) mod noise_len, k] * amplitude_k(n), amplitude_k(n) = the chunk-wise ddsp.core.resample(method='linear') of
 * amplitudes[r, :, k] given as tables lo/hi (int32[N], frame indices) and w (float32[N]).  noise_bands[noise_len, K] is
 * the fixed loopable filtered-noise bank of get_noise_bands (:283-309), built on the host. */
int ddspp_noise_bands(const float* amplitudes, const float* noise_bands, const int* lo, const int* hi, const float* w,
                      float* audio, int R, int T, int K, int N, int noise_len, int shift, hipStream_t stream);

/* stand-in for the reference's unseeded tf.random.uniform([B, N], -1, 1)
 * (filtered_noise_synth.py:39-40): Philox4x32-10, counter = offset + i / 4, key = seed. */
int ddspp_uniform_noise(float* out, size_t n, uint64_t seed, uint64_t offset, hipStream_t stream);
/* the same generator for out[R, n], row r from the counters offset + r * row_stride + i / 4: a stream keyed by
 * (row, position) for piecewise rendering (a piece draws the numbers of its absolute position). */
int ddspp_uniform_noise_rows(float* out, int R, size_t n, uint64_t seed, uint64_t offset, uint64_t row_stride,
                             hipStream_t stream);

/* ---- reverb ---------------------------------------------------------------------------------- */

typedef struct FftConvPlan ddspp_fftconv_plan;

/* ddsp.core.get_fft_size(N, L, power_of_2=True) */
int ddspp_fft_size(int N, int L);
#define DDSPP_DELAY_AUTO (-1)
#define DDSPP_DELAY_AUTO_HALF (-2)

/* ddsp.core.fft_convolve, single IR frame (ddsp.effects.Reverb.get_signal; fdn_reverb.py:407-410).
 * B_ir is B or 1 (a batch-1 IR is shared by all rows). */
int ddspp_fftconv_plan_create(int B, int B_ir, int N, int L, ddspp_fftconv_plan** out_plan);
int ddspp_fftconv_plan_destroy(ddspp_fftconv_plan* plan);
size_t ddspp_fftconv_workspace_bytes(const ddspp_fftconv_plan* plan);
int ddspp_fftconv_fft_size(const ddspp_fftconv_plan* plan);
int ddspp_fftconv_execute(ddspp_fftconv_plan* plan, const float* audio, int audio_stride, const float* ir,
                          float* out, int out_len, int delay, int mask_dry, int add_dry, void* workspace,
                          size_t workspace_bytes, hipStream_t stream);
/* The two halves of ddspp_fftconv_execute, for callers that have the impulse response before the audio: transform_ir
 * (pad, dry mask, R2C of the impulse responses into `workspace`) may be enqueued early, on another stream;
 * execute_prepared (audio R2C, product, C2R, crop + dry) finishes on the same workspace once both are done. */
int ddspp_fftconv_transform_ir(ddspp_fftconv_plan* plan, const float* ir, int mask_dry, void* workspace,
                               size_t workspace_bytes, hipStream_t stream);
int ddspp_fftconv_execute_prepared(ddspp_fftconv_plan* plan, const float* audio, int audio_stride, float* out,
                                   int out_len, int delay, int add_dry, void* workspace, size_t workspace_bytes,
                                   hipStream_t stream);


/* ---- FDN reverb impulse-response generation ---------------------------------- */

/* FeedbackDelayNetwork.get_late_ir up to the irfft -- fdn_reverb.py:178-334, B instruments at once
 * (sub_modules.py:431-446).  gains [B,D], mixing_matrix [D,D], allpass gains/delays [B,D,A],
 * time_rev_0_sec / alpha_tone [B], delay_values [D] -> H [B, freq_points/2+1] complex64.
 * solve: DDSPP_FDN_SOLVE_F64 (0: the per-bin D x D system solved in float64 -- the value the reference's recipe
 * approximates) or DDSPP_FDN_SOLVE_C64_INVERSE (1: tf.linalg.inv + matmuls in complex64 as fdn_reverb.py:314-333
 * writes them; good to cond x 6e-8 near the network's resonances). */
#define DDSPP_FDN_SOLVE_F64 0
#define DDSPP_FDN_SOLVE_C64_INVERSE 1
int ddspp_fdn_transfer(const float* input_gain, const float* output_gain, const float* mixing_matrix,
                       const float* gain_allpass, const float* delays_allpass, const float* time_rev_0_sec,
                       const float* alpha_tone, const float* delay_values, void* H, int B, int D, int A,
                       int freq_points, float sampling_rate, int solve, hipStream_t stream);

/* FeedbackDelayNetwork.get_ir, fdn_reverb.py:354-360: ir[B,L] += zero-padded early_ir[B,E]. */
int ddspp_fdn_add_early(float* ir, const float* early_ir, int B, int L, int E, hipStream_t stream);

/* tf.signal.irfft of any even length n (rocFFT C2R, 1/n scale), batch rows; the spectrum is destroyed. */
typedef struct C2rPlan ddspp_irfft_plan;
int ddspp_irfft_plan_create(int n, int batch, ddspp_irfft_plan** out_plan);
int ddspp_irfft_plan_destroy(ddspp_irfft_plan* plan);
size_t ddspp_irfft_workspace_bytes(const ddspp_irfft_plan* plan);
int ddspp_irfft_execute(ddspp_irfft_plan* plan, void* spectrum, float* signal, void* workspace,
                        size_t workspace_bytes, hipStream_t stream);

/* ---- input edge: MIDI piano roll -> polyphonic conditioning (HOST function, CPU buffers) ------- */

/* MIDIRoll2Conditioning, ddsp_piano/utils/midi_encoders.py:4-104 (called from io_utils.py:118-120): a
 * frame-sequential voice allocator.  roll[n_frames, 88, 2] = (note activity, onset velocity) per key (MIDI
 * 21..108) -> conditioning[n_frames, n_synths, 2] = (activity * pitch, velocity) per channel, a note keeping its
 * channel for as long as it sounds; polyphony[n_frames] = sum of the activities.  The allocator state
 * (assigner, reorder, assigned_pitch -- the reference object's attributes) lives in the handle and carries
 * over from one call to the next, like the reference object's.  All buffers are HOST memory; the roll is not
 * modified (the reference multiplies the caller's activity roll by the pitch in place).
 * Ties: the reference picks the n_synths largest values with an unstable np.argsort; among EQUAL values
 * (in practice: the silent keys, activity 0) this implementation takes the higher keys -- identical output
 * whenever equal-valued keys carry equal velocities, which holds for every roll note_seq produces (a silent
 * key has no onset velocity). */
typedef struct ddspp_midi_state ddspp_midi_state;
ddspp_midi_state* ddspp_midi_conditioning_create(int n_synths);            /* NULL + last_error on bad n_synths */
void ddspp_midi_conditioning_destroy(ddspp_midi_state* state);
int ddspp_midi_conditioning_reset(ddspp_midi_state* state);                /* state of a fresh object, :16-22 */
int ddspp_midi_conditioning_get_state(const ddspp_midi_state* state, int* assigner, int* reorder /* [n_synths] */,
                                      double* assigned_pitch /* [n_synths] */);
int ddspp_midi_conditioning_run_f64(ddspp_midi_state* state, const double* roll, int n_frames, int n_pitches,
                                    double* conditioning, double* polyphony);
int ddspp_midi_conditioning_run_f32(ddspp_midi_state* state, const float* roll, int n_frames, int n_pitches,
                                    float* conditioning, float* polyphony);

/* ---- the whole polyphonic group in one call ---------------------------------------------------------------------
 * processor_group(features, return_outputs_dict=True) -- ddsp_piano/modules/piano_model.py:160 over the DAG of
 * ddsp_piano/modules/polyphonic_dag.py:24-40: P voices x B segments of MultiInharmonic (get_controls + get_signal),
 * FilteredNoise, the MultiAdd chain and Reverb, enqueued as one sequence of kernels (csrc/group.cpp): get_controls
 * over all rows -> compacted oscillator bank -> fused FilteredNoise with voice sums -> add chain -> reverb.
 *
 * ddspp_group_create builds, once per configuration, the small tables the kernels take (uploaded here: the only device
 * memory the library allocates itself) and the reverb's rocFFT plan; ddspp_group_run only launches kernels on `stream`
 * inside the caller's workspace -- no host synchronisation, no allocation: it can be captured in a HIP graph.
 * Rows: every control tensor is [R, T, .] with R = n_segments * n_voices rows, segment major ([B, P]) or voice major
 * ([P, B], what the reference's Parallelizer.unparallelize hands over, sub_modules.py:586-592).
 * Takes the shapes the compacted bank and the even/odd FIR design take (U % 8 == 0, P * S <= 64, K in {32, 64, 96, 128}
 * with the full-length window); FilteredNoise runs fused when its kernel fits the shape, in the two-call form otherwise.
 * The last node is ddsp.effects.Reverb, or (reverb_keep_dry_tap = 1, reverb_add_dry = 0) the apply step of a
 * FeedbackDelayNetwork whose impulse response the caller computed (ddspp_fdn_transfer + ddspp_irfft_* + ddspp_fdn_add_early). */
typedef struct ddspp_group ddspp_group;
typedef struct {
    int n_segments, n_voices, n_frames, n_harmonics, n_substrings, n_bands, upsampling;
    int ir_length;                 /* L of the reverb's impulse response; 0: the DAG has no reverb node */
    int ir_batch;                  /* rows of reverb_ir: 1 (shared by all segments) or n_segments (0 = n_segments) */
    int reverb_add_dry;            /* ddsp.effects.Reverb(add_dry=True) */
    int voice_major;               /* 0: rows are [B, P]; 1: [P, B] */
    float sample_rate, min_frequency;                      /* InHarmonic(sample_rate, min_frequency=20) */
    int scale_kind;                                        /* additive scale_fn: DDSPP_SCALE_* */
    float exponent, max_value, threshold, gain;            /* its parameters (10, 2, 1e-7, 1) */
    int normalize_after_nyquist_cut, normalize_below_nyquist;   /* inharm_synth.py:141-142 */
    int window_size;                                       /* FilteredNoise(window_size=257) */
    int noise_scale_kind;                                  /* FilteredNoise scale_fn (DDSPP_SCALE_*; -1: magnitudes as given) */
    float noise_bias, noise_exponent, noise_max_value, noise_threshold, noise_gain;   /* initial_bias=-5, 10, 2, 1e-7, 1 */
    int delay_compensation;        /* frequency_filter: DDSPP_DELAY_AUTO / DDSPP_DELAY_AUTO_HALF / >= 0 */
    int resize_rule;               /* bilinear rule of ddspp_resample_tables_host (0) */
    uint64_t noise_seed;           /* the library's Philox stream when ddspp_group_run gets noise = NULL */
    int reverb_keep_dry_tap;       /* 0: ddsp.effects.Reverb (ir[:, 0] masked); 1: FeedbackDelayNetwork.get_signal (fdn_reverb.py:407-410:
                                    * the impulse response as it is; use with reverb_add_dry = 0) -- the ENSTDkCl configurations */
    int reserved_;                 /* keep 0 */
} ddspp_group_config;
/* What the reference's outputs dictionary holds besides the audio; every pointer may be NULL (not wanted).  The DAG
 * re-uses one additive and one noise processor for all voices, so the dictionary keeps the LAST voice's stems and
 * conditioned controls (polyphonic_dag.py:28-37). */
typedef struct {
    float* dry;                          /* [B, N]  add/signal: the dry mix the reverb gets */
    float* prev;                         /* [B, N]  add/controls/signal_0: the mix of voices 0 .. P-2 (P > 1) */
    float* additive_last;                /* [B, N]  additive/signal */
    float* noise_last;                   /* [B, N]  noise/signal */
    float* amplitudes_last;              /* [B, T]    additive/controls/amplitudes */
    float* harmonic_distribution_last;   /* [B, T, H] additive/controls/harmonic_distribution */
    float* harmonic_shifts_last;         /* [B, T, H] additive/controls/harmonic_shifts */
    float* magnitudes_last;              /* [B, T, K] noise/controls/magnitudes */
} ddspp_group_outputs;
/* sizeof the two structs above in this build of the library (a binding in another language checks its declaration) */
size_t ddspp_group_config_bytes(void);
size_t ddspp_group_outputs_bytes(void);
int ddspp_group_create(const ddspp_group_config* config, ddspp_group** out_group);
void ddspp_group_destroy(ddspp_group* group);
size_t ddspp_group_workspace_bytes(const ddspp_group* group);
int ddspp_group_n_samples(const ddspp_group* group);      /* N = n_frames * upsampling */
/* amplitudes[R,T], harmonic_distribution[R,T,H], inharm_coef[R,T], f0_hz[R,T,S], magnitudes[R,T,K] (all raw network
 * outputs), reverb_ir[ir_batch, L] (NULL without reverb), noise[R,N] uniform(-1, 1) draws in the rows' order or NULL
 * (the library's Philox stream, one counter step per call) -> audio[B, N].  outputs: NULL = audio only (the voices'
 * noise is summed in registers, no stem is formed); else the dictionary's entries, each optional.  workspace: 256-byte
 * aligned device memory of ddspp_group_workspace_bytes.  One run at a time per group object. */
int ddspp_group_run(ddspp_group* group, const float* amplitudes, const float* harmonic_distribution,
                    const float* inharm_coef, const float* f0_hz, const float* magnitudes, const float* reverb_ir,
                    const float* noise, float* audio, const ddspp_group_outputs* outputs, void* workspace,
                    size_t workspace_bytes, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DDSPP_H_ */
