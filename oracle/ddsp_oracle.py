"""CPU oracle for the DDSP-Piano synthesis hot path (TEST INFRASTRUCTURE ONLY).

    *** PARITY UNPINNED ***
    The reference (lrenault/ddsp-piano @ v2) is Python on top of TensorFlow and
    the un-vendored pip package ``ddsp==3.7.0`` (README.md:11-15).  Neither is
    importable in the build container or on the GPU box, and the reference ships
    no tests / golden vectors.  This file is therefore a
    float32-faithful *restatement*: functions that live in /root/reference cite
    the file:line they follow; functions that live in ``ddsp`` restate the
    published ddsp 3.7.0 algorithm (ddsp/core.py, synths.py, effects.py,
    processors.py) and are anchored on the reference's call sites.  It is pinned
    only by analytic known-answer tests and by scipy/numpy cross-checks
    (tests/test_oracle_kat.py), never by TF outputs.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product (``ddsp_piano_amd``) never
does: it fails loudly when its HIP library is missing.

float32 contract (what "faithful" means here):
  * every elementwise op is a separately rounded IEEE float32 op, in the order
    the reference writes it (no FMA contraction);
  * ``cumsum`` is a sequential float32 scan along time (Eigen scan order on CPU);
  * ``%`` is floormod = fmod + sign fix-up with float32(2*pi) = 6.2831855;
  * FFTs are evaluated in float64 and rounded once to float32 (TF/pocketfft and
    rocFFT both deviate from this by ~1e-7 relative, well inside 1e-4 RMS);
  * transcendental functions (cos, exp, log, tanh, pow) are numpy's float32
    versions; TF's Eigen versions differ in the last ulp.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.fft as sfft

F32 = np.float32
TWO_PI_F32 = F32(2.0 * np.pi)          # float32(6.2831855), what TF makes of `2.0 * pi`

# ----------------------------------------------------------------------------------------------
# The eight details of ddsp 3.7.0 that are RECALLED, not read (SURVEY.md 8(c) "VERIFY" list).  Each is ONE
# switchable entry here; every function below reads it at call time.  The defaults are the builder's
# recollection of ddsp 3.7.0; the alternatives are what a different recollection would give.  A host with
# TensorFlow + ddsp settles them: tests/golden/make_golden.py (DDSP_GOLDEN_BACKEND=tf) regenerates the
# golden vectors from the real library and tests/test_golden_backend.py reports which setting they match.
#   auto_delay      crop_and_compensate_delay, delay_compensation < 0:
#                     'ddsp370'  start = (ir_size - 1) // 2 - 1        'half'  start = ir_size // 2
#   window_crop     apply_window_to_impulse_response when window_size < ir_size:
#                     'ddsp370'  half_idx = (W + 1) // 2, crop [ir_size - half_idx + 2:] ++ [:half_idx + 1]
#                     'centred'  half_idx = W // 2,       crop [ir_size - half_idx:]     ++ [:W - half_idx]
#   resize          core.resample(method='linear') -> tf.compat.v1.image.resize(BILINEAR, align_corners=False):
#                     'legacy'      pos = float32(n) * float32(T / N)                (TF1 kernel, no half-pixel centres)
#                     'half_pixel'  pos = (float32(n) + 0.5) * float32(T / N) - 0.5  (TF2 tf.image.resize)
#   angular_cumsum  core.angular_cumsum:
#                     'ddsp370'    inclusive cumsum in the chunk, chunk offsets shifted right by one chunk
#                     'exclusive'  exclusive cumsum in the chunk (first sample has phase 0)
#   angular_offsets core.angular_cumsum, the running sum of the chunks' end phases that is added to every chunk:
#                     'wrapped'  offsets = tf.cumsum(offsets, axis=1) % (2 pi)   
#                     'plain'    offsets = tf.cumsum(offsets, axis=1)            (the sum grows by up to 2 pi per chunk;
#                                `phase + offsets` is then rounded at the magnitude of the sum: ulp(2 pi n_chunks))
#   angular_wrap    core.angular_cumsum's LAST step (round 6):
#                     'final'  phase = (phase + offsets) % (2 pi): the function returns phases in [0, 2 pi)
#                     'none'   phase + offsets is returned as it is (up to ~2 pi x 500 inside a chunk for a partial near
#                              Nyquist) and tf.cos reduces it by the TRUE 2 pi: float32(2 pi) - 2 pi = 1.75e-7 per turn,
#                              <= 9e-5 rad on the argument of the highest partials' cosine, nothing on the phase state
#   exp_sigmoid     defaults (exponent, max_value, threshold) of core.exp_sigmoid
#   initial_bias    default of synths.FilteredNoise(initial_bias=)
# ----------------------------------------------------------------------------------------------
RECALLED_DEFAULTS = {
    'auto_delay': 'ddsp370',
    'window_crop': 'ddsp370',
    'resize': 'legacy',
    'angular_cumsum': 'ddsp370',
    'angular_offsets': 'wrapped',
    'angular_wrap': 'final',
    'exp_sigmoid': (10.0, 2.0, 1e-7),
    'initial_bias': -5.0,
}
RECALLED_CHOICES = {
    'auto_delay': ('ddsp370', 'half'),
    'window_crop': ('ddsp370', 'centred'),
    'resize': ('legacy', 'half_pixel'),
    'angular_cumsum': ('ddsp370', 'exclusive'),
    'angular_offsets': ('wrapped', 'plain'),
    'angular_wrap': ('final', 'none'),
}
RECALLED = dict(RECALLED_DEFAULTS)


class recalled:
    """``with recalled(auto_delay='half'): ...`` -- evaluate the oracle under another recollection."""

    def __init__(self, **overrides):
        for k, v in overrides.items():
            if k not in RECALLED_DEFAULTS:
                raise KeyError(f'unknown recalled detail {k!r}')
            if k in RECALLED_CHOICES and v not in RECALLED_CHOICES[k]:
                raise ValueError(f'{k} must be one of {RECALLED_CHOICES[k]}, got {v!r}')
        self.overrides = overrides

    def __enter__(self):
        self.saved = dict(RECALLED)
        RECALLED.update(self.overrides)
        return RECALLED

    def __exit__(self, *exc):
        RECALLED.clear()
        RECALLED.update(self.saved)
        return False


def tf_float32(x):
    """ddsp.core.tf_float32: cast/convert to float32."""
    return np.asarray(x, dtype=F32)


# ----------------------------------------------------------------------------------------------
# scale functions / small helpers
# ----------------------------------------------------------------------------------------------
def safe_divide(numerator, denominator, eps=1e-7):
    """ddsp.core.safe_divide: ``a / where(b == 0, eps, b)`` (call sites inharm_synth.py:195,211)."""
    numerator = tf_float32(numerator)
    denominator = tf_float32(denominator)
    safe = np.where(denominator == 0.0, F32(eps), denominator).astype(F32)
    return (numerator / safe).astype(F32)


def _sigmoid_f32(x):
    x = tf_float32(x)
    return (F32(1.0) / (F32(1.0) + np.exp(-x, dtype=F32))).astype(F32)


def exp_sigmoid(x, exponent=None, max_value=None, threshold=None):
    """ddsp.core.exp_sigmoid: ``max_value * sigmoid(x) ** log(exponent) + threshold``.

    Default scale_fn of InHarmonic (inharm_synth.py:149) and FilteredNoise.  The default constants
    (10.0, 2.0, 1e-7) are RECALLED['exp_sigmoid'].
    """
    d = RECALLED['exp_sigmoid']
    exponent = d[0] if exponent is None else exponent
    max_value = d[1] if max_value is None else max_value
    threshold = d[2] if threshold is None else threshold
    x = tf_float32(x)
    p = F32(np.log(F32(exponent)))
    return (F32(max_value) * np.power(_sigmoid_f32(x), p, dtype=F32) + F32(threshold)).astype(F32)


def positive_tanh(x):
    """inharm_synth.py:8-10."""
    x = tf_float32(x)
    return (F32(0.5) * (np.tanh(x, dtype=F32) + F32(1.0))).astype(F32)


def exp_tanh(x, max_value=2.0, exponent=10.0, gain=1.0, threshold=1e-7):
    """inharm_synth.py:13-17."""
    p = F32(np.log(F32(exponent)))
    y = F32(max_value) * np.power(positive_tanh(F32(gain) * tf_float32(x)), p, dtype=F32)
    return (y + F32(threshold)).astype(F32)


def remove_above_nyquist(frequency_envelopes, amplitude_envelopes, sample_rate=16000):
    """ddsp.core.remove_above_nyquist: ``where(f >= sr / 2, 0, a)`` (note ``>=``)."""
    f = tf_float32(frequency_envelopes)
    a = tf_float32(amplitude_envelopes)
    return np.where(f >= F32(sample_rate / 2.0), F32(0.0), a).astype(F32)


def get_harmonic_frequencies(frequencies, n_harmonics):
    """ddsp.core.get_harmonic_frequencies: ``f0 * linspace(1, H, H)``."""
    f_ratios = np.linspace(1.0, float(n_harmonics), int(n_harmonics)).astype(F32)
    return (tf_float32(frequencies) * f_ratios[None, None, :]).astype(F32)


def get_inharmonic_freq(f0_hz, inharm_coef, n_harmonics):
    """inharm_synth.py:20-46."""
    f0_hz = tf_float32(f0_hz)
    inharm_coef = tf_float32(inharm_coef)
    int_multiplier = np.linspace(1.0, float(n_harmonics), int(n_harmonics)).astype(F32)
    int_multiplier = int_multiplier[None, None, :]
    inharm_factor = np.power(int_multiplier, F32(2.0), dtype=F32)          # :37
    inharm_factor = (inharm_factor * inharm_coef).astype(F32) + F32(1.0)   # :38
    inharm_factor = np.sqrt(inharm_factor, dtype=F32)                      # :39
    inharmonic_freq = ((f0_hz * int_multiplier).astype(F32) * inharm_factor).astype(F32)  # :42
    harmonic_shifts = (inharm_factor - F32(1.0)).astype(F32)               # :44
    return inharmonic_freq, harmonic_shifts


# ----------------------------------------------------------------------------------------------
# frame -> sample upsamplers (ddsp.core.resample / upsample_with_windows)
# ----------------------------------------------------------------------------------------------
def hann_window(n, periodic=True):
    """tf.signal.hann_window restated in float32 (window_ops._raised_cosine_window).

    ``even = 1 - n % 2; denom = n + periodic * even - 1; 0.5 - 0.5 * cos(2*pi*i / denom)`` with
    every op in float32.
    """
    n = int(n)
    if n == 1:
        return np.ones([1], F32)
    even = 1 - n % 2
    denom = F32(n + int(periodic) * even - 1)
    count = np.arange(n, dtype=F32)
    cos_arg = ((TWO_PI_F32 * count).astype(F32) / denom).astype(F32)
    return (F32(0.5) - (F32(0.5) * np.cos(cos_arg, dtype=F32)).astype(F32)).astype(F32)


def linear_resample_positions(n_frames, n_timesteps):
    """Legacy (TF1, align_corners=False, no half-pixel centres) bilinear source positions.

    ``scale = float32(T) / float32(N); pos = float32(n) * scale; lo = floor(pos);
    hi = min(ceil(pos), T - 1); w = pos - floor(pos)`` -- all float32 (resize_bilinear CPU kernel,
    reached through ddsp.core.resample -> tf.compat.v1.image.resize).
    """
    scale = F32(n_frames) / F32(n_timesteps)
    if RECALLED['resize'] == 'half_pixel':         # TF2 HalfPixelScaler: (out + 0.5) * scale - 0.5
        pos = (((np.arange(n_timesteps, dtype=F32) + F32(0.5)).astype(F32) * scale).astype(F32) - F32(0.5)).astype(F32)
    else:                                          # TF1 LegacyScaler: out * scale
        pos = (np.arange(n_timesteps, dtype=F32) * scale).astype(F32)
    fl = np.floor(pos).astype(F32)
    # (TF clamps `lower` only from below.  In a file long enough for float32(N - 1) * scale to round up to T -- past
    # 131 072 frames at hop 96 -- its kernel reads one row past the tensor for the very last sample: undefined there;
    # here, and in the library, that sample takes the last row.)
    lo = np.minimum(np.maximum(fl.astype(np.int64), 0), n_frames - 1)
    hi = np.minimum(np.maximum(np.ceil(pos).astype(np.int64), 0), n_frames - 1)
    w = (pos - fl).astype(F32)
    return lo, hi, w


def resample_linear(inputs, n_timesteps):
    """ddsp.core.resample(method='linear'): [B,T,C] -> [B,N,C] (call site inharm_synth.py:117)."""
    x = tf_float32(inputs)
    lo, hi, w = linear_resample_positions(x.shape[1], n_timesteps)
    top = x[:, lo, :]
    bot = x[:, hi, :]
    return (top + ((bot - top).astype(F32) * w[None, :, None]).astype(F32)).astype(F32)


def overlap_and_add(frames, frame_step):
    """tf.signal.overlap_and_add: [..., F, W] -> [..., (F - 1) * step + W]."""
    frames = tf_float32(frames)
    n_frames, frame_len = frames.shape[-2], frames.shape[-1]
    out_len = (n_frames - 1) * frame_step + frame_len
    out = np.zeros(frames.shape[:-2] + (out_len,), F32)
    for f in range(n_frames):
        out[..., f * frame_step: f * frame_step + frame_len] += frames[..., f, :]
    return out


def upsample_with_windows(inputs, n_timesteps, add_endpoint=True):
    """ddsp.core.upsample_with_windows (overlapping Hann windows), literal restatement."""
    x = tf_float32(inputs)
    if x.ndim != 3:
        raise ValueError('Upsample_with_windows() only supports 3 dimensions, not {}.'.format(x.shape))
    if add_endpoint:
        x = np.concatenate([x, x[:, -1:, :]], axis=1)
    n_frames = int(x.shape[1])
    n_intervals = n_frames - 1
    if n_frames >= n_timesteps:
        raise ValueError('Upsample with windows cannot be used for downsampling'
                         'More input frames ({}) than output timesteps ({})'.format(n_frames, n_timesteps))
    if n_timesteps % n_intervals != 0.0:
        raise ValueError('n_timesteps / n_intervals must be an integer')
    hop_size = n_timesteps // n_intervals
    window = hann_window(2 * hop_size)
    xt = np.transpose(x, [0, 2, 1])                       # [B, C, T+1]
    # two-term OLA evaluated half by half (a + b is order independent for two float32 terms)
    out = np.zeros(xt.shape[:2] + (n_frames + 1, hop_size), F32)
    out[:, :, :n_frames, :] += (xt[..., None] * window[:hop_size]).astype(F32)
    out[:, :, 1:, :] += (xt[..., None] * window[hop_size:]).astype(F32)
    out = out.reshape(xt.shape[0], xt.shape[1], -1)
    out = np.transpose(out, [0, 2, 1])
    return np.ascontiguousarray(out[:, hop_size:-hop_size, :])


def resample(inputs, n_timesteps, method='linear', add_endpoint=True):
    """ddsp.core.resample; call sites inharm_synth.py:117-119."""
    x = tf_float32(inputs)
    is_1d = x.ndim == 1
    is_2d = x.ndim == 2
    if is_1d:
        x = x[None, :, None]
    if is_2d:
        x = x[:, :, None]
    if method == 'linear':
        if not add_endpoint:
            raise NotImplementedError('oracle restates align_corners=False only')
        y = resample_linear(x, n_timesteps)
    elif method == 'window':
        y = upsample_with_windows(x, n_timesteps, add_endpoint)
    else:
        raise ValueError('Method ({}) is invalid. Must be one of {}.'.format(
            method, "['nearest', 'linear', 'cubic', 'window']"))
    if is_1d:
        y = y[0, :, 0]
    if is_2d:
        y = y[:, :, 0]
    return y


# ----------------------------------------------------------------------------------------------
# oscillator bank (inharm_synth.py:49-127 + ddsp.core.angular_cumsum)
# ----------------------------------------------------------------------------------------------
def angular_cumsum(angular_frequency, chunk_size=1000):
    """ddsp.core.angular_cumsum: chunked float32 cumsum with 2*pi wrapping."""
    x = tf_float32(angular_frequency)
    n_batch, length = x.shape[0], x.shape[1]
    rest = x.shape[2:]
    remainder = length % chunk_size
    if remainder:
        pad = chunk_size - remainder
        x = np.concatenate([x, np.zeros((n_batch, pad) + rest, F32)], axis=1)
    length_p = x.shape[1]
    n_chunks = length_p // chunk_size
    chunks = x.reshape((n_batch, n_chunks, chunk_size) + rest)
    phase = np.cumsum(chunks, axis=2, dtype=F32)                       # sequential float32 scan
    last = phase[:, :, -1:, ...]
    if RECALLED['angular_cumsum'] == 'exclusive':                      # tf.cumsum(exclusive=True): phase[0] = 0
        phase = np.concatenate([np.zeros_like(phase[:, :, :1]), phase[:, :, :-1]], axis=2)
    offsets = np.mod(last, TWO_PI_F32).astype(F32)
    offsets = np.concatenate([np.zeros_like(offsets[:, :1]), offsets[:, :-1]], axis=1)
    offsets = np.cumsum(offsets, axis=1, dtype=F32)                    # sequential float32 scan over the chunks
    if RECALLED['angular_offsets'] == 'wrapped':
        offsets = np.mod(offsets, TWO_PI_F32).astype(F32)
    phase = (phase + offsets).astype(F32)
    if RECALLED['angular_wrap'] == 'final':
        phase = np.mod(phase, TWO_PI_F32).astype(F32)
    phase = phase.reshape((n_batch, length_p) + rest)
    return phase[:, :length]


def cos_oscillator_bank(frequency_envelopes, amplitude_envelopes, sample_rate=16000,
                        sum_sinusoids=True, use_angular_cumsum=False):
    """inharm_synth.py:49-84."""
    fe = tf_float32(frequency_envelopes)
    ae = remove_above_nyquist(fe, amplitude_envelopes, sample_rate)       # :65-67
    omegas = (fe * TWO_PI_F32).astype(F32)                                # :69
    omegas = (omegas / F32(float(sample_rate))).astype(F32)               # :70
    if use_angular_cumsum:
        phases = angular_cumsum(omegas)                                   # :75
    else:
        phases = np.cumsum(omegas, axis=1, dtype=F32)                     # :77
    wavs = np.cos(phases, dtype=F32)                                      # :80
    audio = (ae * wavs).astype(F32)                                       # :81
    if sum_sinusoids:
        audio = np.sum(audio, axis=-1, dtype=F32)                         # :83
    return audio


def harmonic_synthesis(frequencies, amplitudes, harmonic_shifts=None, harmonic_distribution=None,
                       n_samples=64000, sample_rate=16000, amp_resample_method='window',
                       sum_sinusoids=True, use_angular_cumsum=False):
    """inharm_synth.py:87-127."""
    frequencies = tf_float32(frequencies)
    amplitudes = tf_float32(amplitudes)
    if harmonic_distribution is not None:
        harmonic_distribution = tf_float32(harmonic_distribution)
        n_harmonics = int(harmonic_distribution.shape[-1])
    else:
        n_harmonics = 1
    harmonic_frequencies = get_harmonic_frequencies(frequencies, n_harmonics)          # :106
    if harmonic_shifts is not None:
        harmonic_frequencies = (harmonic_frequencies *
                                (F32(1.0) + tf_float32(harmonic_shifts)).astype(F32)).astype(F32)  # :108
    if harmonic_distribution is not None:
        harmonic_amplitudes = (amplitudes * harmonic_distribution).astype(F32)         # :112
    else:
        harmonic_amplitudes = amplitudes
    frequency_envelopes = resample(harmonic_frequencies, n_samples)                     # :117
    amplitude_envelopes = resample(harmonic_amplitudes, n_samples, method=amp_resample_method)  # :118
    return cos_oscillator_bank(frequency_envelopes, amplitude_envelopes, sample_rate=sample_rate,
                               sum_sinusoids=sum_sinusoids, use_angular_cumsum=use_angular_cumsum)


# ----------------------------------------------------------------------------------------------
# Processor protocol (ddsp.processors)
# ----------------------------------------------------------------------------------------------
class Processor:
    def __init__(self, name, trainable=False):
        self.name = name
        self.trainable = trainable

    def __call__(self, *args, return_outputs_dict=False, **kwargs):
        args = [tf_float32(a) for a in args]
        kwargs = {k: tf_float32(v) for k, v in kwargs.items()}
        controls = self.get_controls(*args, **kwargs)
        signal = self.get_signal(**controls)
        if return_outputs_dict:
            return dict(signal=signal, controls=controls)
        return signal

    def get_controls(self, *args, **kwargs):
        raise NotImplementedError

    def get_signal(self, *args, **kwargs):
        raise NotImplementedError


class InHarmonic(Processor):
    """inharm_synth.py:130-244."""

    def __init__(self, frame_rate=250, sample_rate=16000, min_frequency=20, scale_fn=exp_sigmoid,
                 normalize_after_nyquist_cut=True, normalize_below_nyquist=True, inference=False,
                 name='inharmonic'):
        self.frame_rate = frame_rate
        self.sample_rate = sample_rate
        self.min_frequency = min_frequency
        self.normalize_after_nyquist_cut = normalize_after_nyquist_cut
        self.scale_fn = scale_fn
        self.normalize_below_nyquist = normalize_below_nyquist
        self.inference = inference
        super().__init__(name=name)

    @property
    def upsampling(self):
        return int(self.sample_rate / self.frame_rate)                    # :163-165

    def get_controls(self, amplitudes, harmonic_distribution, inharm_coef, f0_hz):
        amplitudes = tf_float32(amplitudes)
        harmonic_distribution = tf_float32(harmonic_distribution)
        f0_hz = tf_float32(f0_hz)
        inharm_coef = np.maximum(tf_float32(inharm_coef), F32(0.0))       # :183
        if self.scale_fn is not None:
            amplitudes = self.scale_fn(amplitudes)                        # :185
            harmonic_distribution = self.scale_fn(harmonic_distribution)  # :186
        n_harmonics = int(harmonic_distribution.shape[-1])
        inharmonic_freq, harmonic_shifts = get_inharmonic_freq(f0_hz, inharm_coef, n_harmonics)
        if not self.normalize_after_nyquist_cut:                          # :194-198
            harmonic_distribution = safe_divide(
                harmonic_distribution,
                np.sum(harmonic_distribution, axis=-1, keepdims=True, dtype=F32))
        if self.normalize_below_nyquist:                                  # :200-208
            harmonic_distribution = remove_above_nyquist(inharmonic_freq, harmonic_distribution,
                                                         self.sample_rate)
            amplitudes = (amplitudes * (f0_hz > F32(self.min_frequency)).astype(F32)).astype(F32)
        if self.normalize_after_nyquist_cut:                              # :210-214
            harmonic_distribution = safe_divide(
                harmonic_distribution,
                np.sum(harmonic_distribution, axis=-1, keepdims=True, dtype=F32))
        return {'amplitudes': amplitudes, 'harmonic_distribution': harmonic_distribution,
                'harmonic_shifts': harmonic_shifts, 'f0_hz': f0_hz}

    def get_signal(self, amplitudes, harmonic_distribution, harmonic_shifts, f0_hz):
        return harmonic_synthesis(frequencies=f0_hz, amplitudes=amplitudes,
                                  harmonic_shifts=harmonic_shifts,
                                  harmonic_distribution=harmonic_distribution,
                                  n_samples=self.upsampling * f0_hz.shape[1],   # :240
                                  sample_rate=self.sample_rate,
                                  use_angular_cumsum=self.inference)


class MultiInharmonic(InHarmonic):
    """inharm_synth.py:247-293."""

    def __init__(self, name='multi_inharmonic', **kwargs):
        super().__init__(name=name, **kwargs)

    def get_controls(self, amplitudes, harmonic_distribution, inharm_coef, f0_hz):
        f0_hz = tf_float32(f0_hz)
        controls = super().get_controls(amplitudes, harmonic_distribution, inharm_coef,
                                        f0_hz[..., 0:1])                  # :260-265
        controls['f0_hz'] = f0_hz                                         # :267
        controls['amplitudes'] = (controls['amplitudes'] / F32(f0_hz.shape[-1])).astype(F32)  # :269
        return controls

    def get_signal(self, amplitudes, harmonic_distribution, harmonic_shifts, f0_hz):
        n_substrings = f0_hz.shape[-1]
        audio = super().get_signal(amplitudes, harmonic_distribution, harmonic_shifts,
                                   f0_hz[..., 0:1])                       # :279-284
        for substring in range(1, n_substrings):                          # :286-292
            audio = (audio + super().get_signal(amplitudes, harmonic_distribution, harmonic_shifts,
                                                f0_hz[..., substring:substring + 1])).astype(F32)
        return audio


class MultiAdd(Processor):
    """inharm_synth.py:296-309: python ``sum`` of the signals (0 + s0 + s1 + ...)."""

    def __init__(self, name='add'):
        super().__init__(name=name)

    def get_controls(self, *signals):
        return {f'signal_{i}': s for i, s in enumerate(signals)}

    def get_signal(self, **signals):
        vals = list(signals.values())
        out = tf_float32(vals[0])
        for v in vals[1:]:
            out = (out + tf_float32(v)).astype(F32)
        return out


def multi_add(signals):
    """MultiAdd()(s0, s1, ...) (inharm_synth.py:296-309): python `sum` over the controls' values, left to right."""
    return MultiAdd()(*signals)


class Add(Processor):
    """ddsp.processors.Add (used by default_model.py:56-74)."""

    def __init__(self, name='add'):
        super().__init__(name=name)

    def get_controls(self, signal_one, signal_two):
        return {'signal_one': signal_one, 'signal_two': signal_two}

    def get_signal(self, signal_one, signal_two):
        return (tf_float32(signal_one) + tf_float32(signal_two)).astype(F32)


# ----------------------------------------------------------------------------------------------
# FilteredNoise (ddsp.synths.FilteredNoise / ddsp.core.frequency_filter)
# ----------------------------------------------------------------------------------------------
def get_fft_size(frame_size, ir_size, power_of_2=True):
    """ddsp.core.get_fft_size."""
    convolved_frame_size = ir_size + frame_size - 1
    if power_of_2:
        return int(2 ** np.ceil(np.log2(convolved_frame_size)))
    return int(convolved_frame_size)


def apply_window_to_impulse_response(impulse_response, window_size=0, causal=False):
    """ddsp.core.apply_window_to_impulse_response (zero-phase in, causal linear-phase out)."""
    ir = tf_float32(impulse_response)
    if causal:
        ir = np.fft.fftshift(ir, axes=-1)
    ir_size = int(ir.shape[-1])
    if (window_size <= 0) or (window_size > ir_size):
        window_size = ir_size
    window = hann_window(window_size)
    padding = ir_size - window_size
    centred = RECALLED['window_crop'] == 'centred'
    if padding > 0:
        half_idx = window_size // 2 if centred else (window_size + 1) // 2
        window = np.concatenate([window[half_idx:], np.zeros([padding], F32), window[:half_idx]], axis=0)
    else:
        window = np.fft.fftshift(window, axes=-1)
    ir = (window * ir).astype(F32)
    if padding > 0:
        if centred:
            first_half_start = ir_size - half_idx
            second_half_end = window_size - half_idx
        else:
            first_half_start = (ir_size - (half_idx - 1)) + 1
            second_half_end = half_idx + 1
        ir = np.concatenate([ir[..., first_half_start:], ir[..., :second_half_end]], axis=-1)
    else:
        ir = np.fft.fftshift(ir, axes=-1)
    return np.ascontiguousarray(ir)


def frequency_impulse_response(magnitudes, window_size=0):
    """ddsp.core.frequency_impulse_response: irfft(complex(mag, 0)) then window."""
    mags = tf_float32(magnitudes)
    ir = sfft.irfft(mags.astype(np.float64), axis=-1).astype(F32)      # length 2 (K - 1)
    return apply_window_to_impulse_response(ir, window_size)


def crop_and_compensate_delay(audio, audio_size, ir_size, padding, delay_compensation):
    """ddsp.core.crop_and_compensate_delay."""
    if padding == 'valid':
        crop_size = ir_size + audio_size - 1
    elif padding == 'same':
        crop_size = audio_size
    else:
        raise ValueError('Padding must be \'valid\' or \'same\', instead of {}.'.format(padding))
    total_size = int(audio.shape[-1])
    crop = total_size - crop_size
    if delay_compensation < 0:
        start = ir_size // 2 if RECALLED['auto_delay'] == 'half' else (ir_size - 1) // 2 - 1
    else:
        start = delay_compensation
    end = crop - start
    return audio[:, start:total_size - end] if end != 0 else audio[:, start:0]


def fft_convolve(audio, impulse_response, padding='same', delay_compensation=-1):
    """ddsp.core.fft_convolve (call site fdn_reverb.py:409; used by Reverb and frequency_filter)."""
    audio = tf_float32(audio)
    ir = tf_float32(impulse_response)
    if ir.ndim == 2:
        ir = ir[:, None, :]
    batch_size_ir, n_ir_frames, ir_size = ir.shape
    batch_size, audio_size = audio.shape
    if batch_size_ir == 1 and batch_size > 1:
        ir = np.tile(ir, [batch_size, 1, 1])
        batch_size_ir = batch_size
    if batch_size != batch_size_ir:
        raise ValueError('Batch size of audio ({}) and impulse response ({}) must be the same.'.format(
            batch_size, batch_size_ir))
    frame_size = int(np.ceil(audio_size / n_ir_frames))
    hop_size = frame_size
    n_audio_frames = int(np.ceil(audio_size / hop_size))                 # tf.signal.frame(pad_end=True)
    if n_audio_frames != n_ir_frames:
        raise ValueError('Number of Audio frames ({}) and impulse response frames ({}) do not match. '
                         'For small hop size = ceil(audio_size / n_ir_frames), number of impulse '
                         'response frames must be a multiple of the audio size.'.format(
                             n_audio_frames, n_ir_frames))
    padded = np.zeros([batch_size, n_audio_frames * hop_size], F32)
    padded[:, :audio_size] = audio
    audio_frames = padded.reshape(batch_size, n_audio_frames, frame_size)
    fft_size = get_fft_size(frame_size, ir_size, power_of_2=True)
    audio_fft = sfft.rfft(audio_frames.astype(np.float64), fft_size, axis=-1)
    ir_fft = sfft.rfft(ir.astype(np.float64), fft_size, axis=-1)
    audio_frames_out = sfft.irfft(audio_fft * ir_fft, fft_size, axis=-1).astype(F32)
    audio_out = overlap_and_add(audio_frames_out, hop_size)
    return np.ascontiguousarray(
        crop_and_compensate_delay(audio_out, audio_size, ir_size, padding, delay_compensation))


def frequency_filter(audio, magnitudes, window_size=0, padding='same'):
    """ddsp.core.frequency_filter (call site filtered_noise_synth.py:41-42)."""
    impulse_response = frequency_impulse_response(magnitudes, window_size=window_size)
    return fft_convolve(audio, impulse_response, padding=padding)


class FilteredNoise(Processor):
    """ddsp.synths.FilteredNoise + DynamicSizeFilteredNoise (filtered_noise_synth.py:12-42).

    The reference draws ``tf.random.uniform([B, U*T], -1, 1)`` unseeded (:39-40); the oracle takes
    the noise as an explicit argument so that the deterministic part can be compared.
    """

    def __init__(self, frame_rate=250, sample_rate=16000, window_size=257, scale_fn=exp_sigmoid,
                 initial_bias=None, name='filtered_noise'):
        super().__init__(name=name)
        self.frame_rate = frame_rate
        self.sample_rate = sample_rate
        self.window_size = window_size
        self.scale_fn = scale_fn
        self.initial_bias = RECALLED['initial_bias'] if initial_bias is None else initial_bias

    @property
    def upsampling(self):
        return int(self.sample_rate / self.frame_rate)

    def get_controls(self, magnitudes):
        magnitudes = tf_float32(magnitudes)
        if self.scale_fn is not None:
            magnitudes = self.scale_fn((magnitudes + F32(self.initial_bias)).astype(F32))
        return {'magnitudes': magnitudes}

    def get_signal(self, magnitudes, noise=None):
        batch_size, n_frames = magnitudes.shape[0], magnitudes.shape[1]
        n_samples = self.upsampling * n_frames
        if noise is None:
            raise ValueError('oracle FilteredNoise needs the noise tensor [B, U*T] explicitly')
        noise = tf_float32(noise)
        assert noise.shape == (batch_size, n_samples)
        return frequency_filter(noise, magnitudes, window_size=self.window_size)


# ----------------------------------------------------------------------------------------------
# Reverb (ddsp.effects.Reverb) and FDN apply step (fdn_reverb.py:407-410)
# ----------------------------------------------------------------------------------------------
class Reverb(Processor):
    def __init__(self, trainable=False, reverb_length=48000, add_dry=True, name='reverb'):
        super().__init__(name=name, trainable=trainable)
        self._reverb_length = reverb_length
        self._add_dry = add_dry

    @staticmethod
    def _mask_dry_ir(ir):
        ir = tf_float32(ir)
        if ir.ndim == 1:
            ir = ir[None, :]
        if ir.ndim == 3:
            ir = ir[:, :, 0]
        dry_mask = np.zeros([ir.shape[0], 1], F32)
        return np.concatenate([dry_mask, ir[:, 1:]], axis=1)

    def get_controls(self, audio, ir=None):
        if self.trainable:
            raise NotImplementedError('trainable Reverb is out of scope (SURVEY.md 8a-11)')
        if ir is None:
            raise ValueError('Must provide "ir" tensor if Reverb trainable=False.')
        return {'audio': audio, 'ir': ir}

    def get_signal(self, audio, ir):
        audio, ir = tf_float32(audio), tf_float32(ir)
        ir = self._mask_dry_ir(ir)
        wet = fft_convolve(audio, ir, padding='same', delay_compensation=0)
        return (wet + audio).astype(F32) if self._add_dry else wet


def fdn_get_signal(audio, ir):
    """FeedbackDelayNetwork.get_signal, fdn_reverb.py:407-410: no dry mask, no add-dry."""
    ir = tf_float32(ir)[None, :]
    return fft_convolve(audio, ir, delay_compensation=0)


# ----------------------------------------------------------------------------------------------
# ProcessorGroup / DAG (ddsp.processors.ProcessorGroup, ddsp.dags.DAGLayer)
# ----------------------------------------------------------------------------------------------
def _nested_lookup(key, d):
    out = d
    for k in key.split('/'):
        out = out[k]
    return out


class ProcessorGroup:
    def __init__(self, dag, name='processor_group'):
        self.dag = list(dag)
        self.name = name
        self.processors = []
        for node in self.dag:
            p = node[0]
            if p not in self.processors:
                self.processors.append(p)

    def __call__(self, inputs, return_outputs_dict=False, extra_kwargs=None):
        """extra_kwargs: {processor_name: [kwargs per call]} -- oracle-only hook to pass noise."""
        outputs = {'inputs': inputs}
        outputs.update(inputs)
        counters = {}
        module_outputs = None
        for node in self.dag:
            processor, input_keys = node[0], node[1]
            args = [_nested_lookup(k, outputs) for k in input_keys]
            kw = {}
            if extra_kwargs and processor.name in extra_kwargs:
                i = counters.get(processor.name, 0)
                kw = extra_kwargs[processor.name][i]
                counters[processor.name] = i + 1
            controls = processor.get_controls(*[tf_float32(a) for a in args])
            signal = processor.get_signal(**controls, **kw)
            module_outputs = {'signal': signal, 'controls': controls}
            outputs[processor.name] = module_outputs
        outputs['out'] = module_outputs
        if return_outputs_dict:
            return {'signal': module_outputs['signal'], 'controls': outputs}
        return module_outputs['signal']


def polyphonic_dag(additive, noise, reverb=None,
                   additive_controls=('amps', 'harmonic_distribution', 'f0_hz'),
                   noise_controls=('noise_magnitudes',), reverb_controls=(), n_synths=16):
    """polyphonic_dag.py:5-42."""
    add = MultiAdd(name='add')
    dag = [(additive, [c + '_0' for c in additive_controls]),
           (noise, [c + '_0' for c in noise_controls]),
           (add, [noise.name + '/signal', additive.name + '/signal'])]
    for i in range(1, n_synths):
        dag.append((additive, [c + f'_{i}' for c in additive_controls]))
        dag.append((noise, [c + f'_{i}' for c in noise_controls]))
        dag.append((add, ['add/signal', noise.name + '/signal', additive.name + '/signal']))
    if reverb is not None:
        dag.append((reverb, ['add/signal'] + list(reverb_controls)))
    return dag


def midi_to_hz(notes):
    """ddsp.core.midi_to_hz."""
    return 440.0 * (2.0 ** ((np.asarray(notes, np.float64) - 69.0) / 12.0))


# ----------------------------------------------------------------------------------------------
# SURVEY.md 8f-1 ("next" row): FDN impulse-response generation, fdn_reverb.py:178-360
# (complex64 restatement; used by FeedbackDelayNetwork.get_controls and by
# MultiInstrumentFeedbackDelayReverb.call, sub_modules.py:431-446)
# ----------------------------------------------------------------------------------------------
C64 = np.complex64
FDN_DELAY_VALUES = np.asarray([233, 311, 421, 461, 587, 613, 789, 891], F32)            # fdn_reverb.py:96
FDN_DELAYS_ALLPASS = np.asarray([[131, 151, 337, 353], [103, 173, 331, 373], [89, 181, 307, 401],
                                 [79, 197, 281, 419], [61, 211, 257, 431], [47, 229, 251, 443],
                                 [81, 189, 287, 407], [91, 203, 321, 377]], F32)         # fdn_reverb.py:102-113


def fdn_mixing_matrix(delay_lines):
    """fdn_reverb.py:118-120: Householder-style mixing matrix -I + 0.5 * ones."""
    return (-1.0 * np.eye(delay_lines, dtype=F32) + F32(0.5) * np.ones([delay_lines, delay_lines], F32)).astype(F32)


def fdn_get_late_ir(input_gain, output_gain, mixing_matrix, gain_allpass, delays_allpass, time_rev_0_sec,
                    alpha_tone, delay_values=FDN_DELAY_VALUES, sampling_rate=16000.0, exact_solve=False):
    """FeedbackDelayNetwork.get_late_ir, fdn_reverb.py:178-334, one instrument.

    exact_solve=False follows the reference: the 8x8 systems are inverted in complex64, which is only
    good to cond(I - F D) x 6e-8 near the network's resonances.  exact_solve=True keeps every float32 /
    complex64 quantity the reference forms (frequencies, delays, filter and all-pass transfers) but
    assembles, inverts and projects in complex128 -- the value both implementations approximate.
    """
    sr = F32(sampling_rate)
    freq_points = int(2 * sr)                                            # :81
    nb = freq_points // 2 + 1
    delay_values = tf_float32(delay_values)
    n_lines = delay_values.shape[0]
    input_gain = tf_float32(input_gain).astype(C64)
    output_gain = tf_float32(output_gain).astype(C64)
    mixing = tf_float32(mixing_matrix).astype(C64)
    # Transcendentals (exp(j phi), 10 ** x) are evaluated in float64 and rounded once to float32 -- the correctly
    # rounded float32 value that any float32 libm (TF's, numpy's, the GPU's) approximates to an ulp or two.  Their
    # ARGUMENTS are formed in float32 exactly as the reference forms them (wk * delay, -3 * delay / T60, ...).
    def expj(phi32):                       # exp(1j * phi) for float32 phi
        phi = np.asarray(phi32, F32).astype(np.float64)
        return (np.cos(phi).astype(F32) + 1j * np.sin(phi).astype(F32)).astype(C64)

    def pow10(x32):
        return np.power(10.0, np.asarray(x32, F32).astype(np.float64)).astype(F32)

    wk32 = ((F32(2 * np.pi) * np.arange(nb, dtype=F32)).astype(F32) / F32(freq_points)).astype(F32)   # :234-239
    wk = wk32.astype(C64)
    z_d = np.stack([expj(-(wk32 * np.floor(delay_values[d])).astype(F32)) for d in range(n_lines)], axis=1)   # :241-250
    d_eta = (delay_values - np.floor(delay_values)).astype(F32).astype(C64)
    eta = ((C64(1) - d_eta) / (C64(1) + d_eta)).astype(C64)               # :253-254
    ez = expj(-wk32)
    allpass_interp = np.stack([((eta[d] + ez) / (C64(1) + eta[d] * ez)).astype(C64) for d in range(n_lines)],
                              axis=1)                                     # :255-261
    dd = (z_d * allpass_interp).astype(C64)                               # diagonal of diag_delay_matrix :263
    delays_allpass = tf_float32(delays_allpass)
    gain_allpass = tf_float32(gain_allpass)
    delay_sec = ((delay_values + np.sum(delays_allpass, axis=-1, dtype=F32)).astype(F32) / sr).astype(F32)   # :265-268
    t0 = F32(time_rev_0_sec)
    al = F32(alpha_tone)
    k = pow10((F32(-3) * delay_sec / t0).astype(F32))                     # :272
    kpi = pow10((F32(-3) * delay_sec / (al * t0)).astype(F32))           # :274-277
    g = (F32(2) * k * kpi / (k + kpi)).astype(F32)
    p = ((k - kpi) / (k + kpi)).astype(F32)
    filt = (g.astype(C64)[None, :] / (C64(1) - p.astype(C64)[None, :] * ez[:, None] + C64(1e-8))).astype(C64)  # :289-291
    z_delays = expj((wk32[:, None, None] * delays_allpass[None]).astype(F32))                                   # :302
    ga = gain_allpass.astype(C64)[None]
    allpass_transfer = np.prod(((C64(1) + ga * z_delays) / (ga + z_delays)).astype(C64), axis=-1).astype(C64)  # :305-308
    feedback = (filt[:, :, None] * mixing[None, :, :] * allpass_transfer[:, None, :]).astype(C64)               # :314-316
    if exact_solve:
        c128 = np.complex128
        fb = filt.astype(c128)[:, :, None] * mixing.astype(c128)[None] * allpass_transfer.astype(c128)[:, None, :]
        a = np.eye(n_lines, dtype=c128)[None] - fb * dd.astype(c128)[:, None, :]
        x = np.linalg.solve(a, np.broadcast_to(input_gain.astype(c128)[None, :, None], (nb, n_lines, 1)))[..., 0]
        h = np.sum(output_gain.astype(c128)[None] * dd.astype(c128) * x, axis=1).astype(C64)
        return np.fft.irfft(h.astype(np.complex128)).astype(F32)
    eye = np.eye(n_lines, dtype=C64)[None]
    a = (eye - feedback * dd[:, None, :]).astype(C64)                     # I - F @ diag(dd)
    inv = np.linalg.inv(a).astype(C64)
    m = (dd[:, :, None] * inv).astype(C64)                                # diag(dd) @ inv
    h = np.einsum('i,nij,j->n', output_gain, m, input_gain).astype(C64)   # :323-333
    return np.fft.irfft(h.astype(np.complex128)).astype(F32)


def fdn_get_ir(input_gain, output_gain, gain_allpass, delays_allpass, time_rev_0_sec, alpha_tone, early_ir,
               delay_values=FDN_DELAY_VALUES, sampling_rate=16000.0, exact_solve=False):
    """FeedbackDelayNetwork.get_ir, fdn_reverb.py:336-360."""
    n_lines = tf_float32(delay_values).shape[0]
    late = fdn_get_late_ir(input_gain, output_gain, fdn_mixing_matrix(n_lines), gain_allpass, delays_allpass,
                           time_rev_0_sec, alpha_tone, delay_values, sampling_rate, exact_solve)
    early = np.squeeze(tf_float32(early_ir))
    if late.shape[0] > early.shape[0]:
        early = np.pad(early, [[0, late.shape[0] - early.shape[0]]])
    return (early[:late.shape[0]] + late).astype(F32)


def reshape_embedding(embedding, splits=4):
    """MultiInstrumentFeedbackDelayReverb.reshape_embedding, sub_modules.py:427-429:
    tf.stack(tf.split(embedding, splits, axis=-1), axis=-1): [..., splits * D] -> [..., D, splits]."""
    return np.stack(np.split(np.asarray(embedding), splits, axis=-1), axis=-1)


class MultiInstrumentFeedbackDelayReverb:
    """sub_modules.py:368-446 with the Embedding tables given as arrays [n_instruments, width] (names as the layer's
    attributes without the underscore).  call(piano_model [B, 1]) -> reverb_ir [B, 2 * sample_rate]."""

    def __init__(self, tables, n_instruments, sample_rate=16000, exact_solve=False):
        self.tables = {k: tf_float32(v) for k, v in tables.items()}
        self.n_instruments = n_instruments
        self.sample_rate = sample_rate
        self.exact_solve = exact_solve

    def controls(self, piano_model):
        pm = np.asarray(piano_model)
        if self.n_instruments == 1:                                          # :432-433
            pm = np.zeros_like(pm, dtype=np.int32)
        pm = pm[..., 0]                                                      # :434
        t = self.tables
        return {'input_gain': t['input_gain'][pm], 'output_gain': t['output_gain'][pm],
                'gain_allpass': reshape_embedding(t['gain_allpass'][pm]),
                'delays_allpass': reshape_embedding(t['delays_allpass'][pm]),
                'time_rev_0_sec': np.maximum(t['time_rev_0_sec'][pm], F32(0)),               # tf.nn.relu, :440
                'alpha_tone': (F32(1) / (F32(1) + np.exp(-t['alpha_tone'][pm]))).astype(F32),  # tf.math.sigmoid, :441
                'early_ir': t['early_ir'][pm]}

    def __call__(self, piano_model):
        c = self.controls(piano_model)
        b = c['input_gain'].shape[0]
        return np.stack([fdn_get_ir(c['input_gain'][i], c['output_gain'][i], c['gain_allpass'][i], c['delays_allpass'][i],
                                    c['time_rev_0_sec'][i, 0], c['alpha_tone'][i, 0], c['early_ir'][i],
                                    sampling_rate=self.sample_rate, exact_solve=self.exact_solve) for i in range(b)])


class MultiInstrumentReverb:
    """sub_modules.py:302-365: reverb_dict [n_instruments, reverb_length]; exponential decay mask at inference."""

    def __init__(self, reverb_dict, n_instruments, inference=False):
        self.reverb_dict = tf_float32(reverb_dict)
        self.n_instruments = n_instruments
        self.inference = inference

    def __call__(self, piano_model, decay_exponent=4., decay_start=16000):
        pm = np.asarray(piano_model)
        if self.n_instruments == 1:
            pm = np.zeros_like(pm, dtype=np.int32)
        ir = self.reverb_dict[pm]
        if ir.ndim == 3:
            ir = ir[:, 0]
        if self.inference:                                                   # :339-349
            n = self.reverb_dict.shape[1]
            time = np.linspace(0.0, 1.0, n - decay_start).astype(F32)
            mask = np.concatenate([np.ones(decay_start, F32), np.exp(F32(-decay_exponent) * time).astype(F32)])
            ir = (ir * mask[None, :]).astype(F32)
        return ir


# ----------------------------------------------------------------------------------------------
# SURVEY.md 8f-3 ("next" row): SurrogateAdditive, ddsp_piano/modules/surrogate_synth.py
# ----------------------------------------------------------------------------------------------
def surrogate_harmonic_synthesis(frequencies, amplitudes, decays=None, decay_time=None, harmonic_shifts=None,
                                 harmonic_distribution=None, upsampling=64, sample_rate=16000,
                                 amp_resample_method='window', use_angular_cumsum=False):
    """surrogate_synth.py:11-104."""
    frequencies, amplitudes = tf_float32(frequencies), tf_float32(amplitudes)
    n_frames = int(frequencies.shape[1])
    n_samples = upsampling * n_frames
    if harmonic_distribution is not None:
        harmonic_distribution = tf_float32(harmonic_distribution)
        n_harmonics = int(harmonic_distribution.shape[-1])
    elif harmonic_shifts is not None:
        n_harmonics = int(tf_float32(harmonic_shifts).shape[-1])
    else:
        n_harmonics = 1
    harmonic_frequencies = get_harmonic_frequencies(frequencies, n_harmonics)            # :59-60
    if harmonic_shifts is not None:
        harmonic_frequencies = (harmonic_frequencies * (F32(1.0) + tf_float32(harmonic_shifts)).astype(F32)).astype(F32)
    harmonic_amplitudes = (amplitudes * harmonic_distribution).astype(F32) if harmonic_distribution is not None \
        else amplitudes                                                                  # :65-68
    frequency_envelopes = resample(harmonic_frequencies, n_samples)                       # :71
    amplitude_envelopes = resample(harmonic_amplitudes, n_samples, method=amp_resample_method)   # :72-73
    if decays is not None and decay_time is not None:
        decays, decay_time = tf_float32(decays), tf_float32(decay_time)
        decay_env = np.repeat(decays, upsampling, axis=1)                                 # :82
        t_up = (np.repeat(decay_time, upsampling, axis=1) * F32(upsampling)).astype(F32)  # :83-84
        rng = np.tile(np.arange(upsampling, dtype=F32), n_frames)[None, :, None]          # :86-89
        t_up = (t_up + rng).astype(F32)
        decay_env = np.power(np.abs(decay_env), t_up, dtype=F32)                          # :91-92
        amplitude_envelopes = (amplitude_envelopes * decay_env).astype(F32)               # :95
    return cos_oscillator_bank(frequency_envelopes, amplitude_envelopes, sample_rate=sample_rate,
                               use_angular_cumsum=use_angular_cumsum)


class SurrogateAdditive(Processor):
    """surrogate_synth.py:107-214."""

    def __init__(self, frame_rate=250, sample_rate=16000, min_frequency=20, normalize_harm_distribution=True,
                 scale_fn=exp_sigmoid, normalize_below_nyquist=True, inference=False, name='inharmonic'):
        super().__init__(name=name)
        self.frame_rate, self.sample_rate, self.min_frequency = frame_rate, sample_rate, min_frequency
        self.normalize_harm_distribution = normalize_harm_distribution
        self.scale_fn = scale_fn
        self.normalize_below_nyquist = normalize_below_nyquist
        self.inference = inference

    def get_controls(self, amplitudes, decays, decay_time, harmonic_distribution, inharm_coef, f0_hz):
        amplitudes, harmonic_distribution = tf_float32(amplitudes), tf_float32(harmonic_distribution)
        f0_hz = tf_float32(f0_hz)
        if self.scale_fn is not None:
            amplitudes = self.scale_fn(amplitudes)
            harmonic_distribution = self.scale_fn(harmonic_distribution)
        inharm_coef = np.maximum(tf_float32(inharm_coef), F32(0.0))
        n_harmonics = int(harmonic_distribution.shape[-1])
        inharmonic_freq, harmonic_shifts = get_inharmonic_freq(f0_hz, inharm_coef, n_harmonics)
        if decays is not None:
            decays = np.maximum(np.minimum(tf_float32(decays), F32(1.0)), F32(1e-5))      # :164-165
            decays = np.where(inharmonic_freq >= F32(self.sample_rate / 2.0), F32(1.0), decays).astype(F32)
        if self.normalize_below_nyquist:
            harmonic_distribution = remove_above_nyquist(inharmonic_freq, harmonic_distribution, self.sample_rate)
            amplitudes = (amplitudes * (f0_hz > F32(self.min_frequency)).astype(F32)).astype(F32)
        if self.normalize_harm_distribution:
            harmonic_distribution = safe_divide(harmonic_distribution,
                                                np.sum(harmonic_distribution, axis=-1, keepdims=True, dtype=F32))
        return {'amplitudes': amplitudes, 'decays': decays, 'decay_time': decay_time,
                'harmonic_distribution': harmonic_distribution, 'harmonic_shifts': harmonic_shifts, 'f0_hz': f0_hz}

    def get_signal(self, amplitudes, decays, decay_time, harmonic_distribution, harmonic_shifts, f0_hz):
        return surrogate_harmonic_synthesis(frequencies=f0_hz, amplitudes=amplitudes, decays=decays,
                                            decay_time=decay_time, harmonic_shifts=harmonic_shifts,
                                            harmonic_distribution=harmonic_distribution,
                                            upsampling=int(self.sample_rate / self.frame_rate),
                                            sample_rate=self.sample_rate, use_angular_cumsum=self.inference)


# ---------------------------------------------------------------------------------------------
# SURVEY.md 8f-3, second half: NoiseBandNetSynth + FilterBank, ddsp_piano/modules/filtered_noise_synth.py:51-317.
# scipy.signal (kaiserord, firwin) is the same third-party code the reference calls.  Two TF random draws cannot be
# restated (a seeded tf.random.uniform stream for the band phases :289-291, an unseeded one for the per-call roll
# :226-230): both are explicit arguments here; the parity tests hand the same values to the HIP path.
# ---------------------------------------------------------------------------------------------
def nbn_frequency_bands(n_filters_linear, n_filters_log, linear_min_f, linear_max_f_cutoff_fs, sample_rate):
    """FilterBank.get_frequency_bands (:103-113) with get_linear_bands (:86-90) and get_log_bands (:92-101)."""
    linear_max_f = (sample_rate / 2) / linear_max_f_cutoff_fs
    linear_bands = np.linspace(linear_min_f, linear_max_f, n_filters_linear)
    linear_bands = np.vstack((linear_bands[:-1], linear_bands[1:])).T
    if linear_max_f_cutoff_fs == 1:
        # :108-109 `return linear_center_f` -- an undefined name: the reference raises NameError on this branch.
        # The restatement returns the linear bands (the evident intent); no shipped configuration reaches it.
        return linear_bands
    log_bands = np.geomspace(start=linear_max_f, stop=sample_rate / 2, num=n_filters_log, endpoint=False)
    log_bands = np.vstack((log_bands[:-1], log_bands[1:])).T
    return np.concatenate((linear_bands, log_bands))


def nbn_filter(cutoff, sample_rate, attenuation, pass_zero, transition_bandwidth=0.2, scale=True):
    """FilterBank.get_filter (:121-133)."""
    from scipy import signal
    if isinstance(cutoff, np.ndarray):
        bandwidth = abs(cutoff[1] - cutoff[0])
    elif pass_zero is True:
        bandwidth = cutoff
    else:
        bandwidth = abs((sample_rate / 2) - cutoff)
    width = (bandwidth / (sample_rate / 2)) * transition_bandwidth
    n_taps, beta = signal.kaiserord(ripple=attenuation, width=width)
    n_taps = 2 * (n_taps // 2) + 1
    return signal.firwin(numtaps=n_taps, cutoff=cutoff, window=('kaiser', beta), scale=scale, fs=sample_rate,
                         pass_zero=pass_zero)


def nbn_filterbank(n_band, sample_rate, attenuation=50, linear_min_f=20, linear_max_f_cutoff_fs=4):
    """FilterBank.build_filterbank (:135-158) for NoiseBandNetSynth.build's arguments (:196-201)."""
    bands = nbn_frequency_bands(n_band // 2, n_band // 2, linear_min_f, linear_max_f_cutoff_fs, sample_rate)
    filters = []
    for i in range(bands.shape[0]):
        if i == 0:
            filters.append(nbn_filter(bands[i, 0], sample_rate, attenuation, True))
        filters.append(nbn_filter(bands[i], sample_rate, attenuation, False))
        if i == bands.shape[0] - 1:
            filters.append(nbn_filter(bands[i, -1], sample_rate, attenuation, False))
    return filters


def nbn_noise_bands(filters, min_noise_len, normalize, phase_noise):
    """get_noise_bands (:283-309): [1, noise_len, n_band] float32 and noise_len; phase_noise [n_band, noise_len/2+1]."""
    max_len = max(len(h) for h in filters)
    noise_len = int(math.pow(2, math.ceil(math.log(max_len) / math.log(2)))) if max_len > min_noise_len else min_noise_len
    padded = np.array([np.pad(h, (noise_len - len(h), 0)) for h in filters]).astype(F32)          # pad_filters :271-274
    magnitude = np.abs(sfft.rfft(padded.astype(np.float64), axis=-1)).astype(F32).astype(C64)    # :277-280
    phase = np.exp(1j * np.asarray(phase_noise, F32).astype(np.float64)).astype(C64)             # :292
    phase[:, 0] = 0                                                                              # :293-297
    phase[:, -1] = 0
    bands = sfft.irfft((magnitude * phase).astype(np.complex128), n=noise_len, axis=-1).astype(F32)   # :299-300
    if normalize:
        bands = (bands / np.max(np.abs(bands))).astype(F32)                                      # :301-302
    return np.transpose(bands[None], [0, 2, 1]), noise_len                                        # :303-306


def nbn_get_signal(amplitudes, noise_bands, noise_len, upsampling, shift):
    """NoiseBandNetSynth.get_signal (:213-262), the loop written as the reference writes it."""
    amplitudes = tf_float32(amplitudes)
    frame_len = int(noise_len / upsampling)
    n_frames = math.ceil(amplitudes.shape[1] / frame_len)
    nb = np.roll(noise_bands, shift, axis=1)                                                     # :224-231
    n_samples = amplitudes.shape[1] * upsampling
    if amplitudes.shape[1] / frame_len < 1:                                                      # :234-238
        up = resample(amplitudes, n_samples)
        return np.sum((nb[:, :n_samples, :] * up).astype(F32), axis=-1, dtype=F32)
    signal = None
    for i in range(n_frames):
        if i == 0:
            up = resample(amplitudes[:, :frame_len], frame_len * upsampling)
            signal = np.sum((nb * up).astype(F32), axis=-1, dtype=F32)
        elif i == n_frames - 1:
            up = resample(amplitudes[:, i * frame_len:], frame_len * upsampling)
            signal = np.concatenate([signal, np.sum((nb[:, :up.shape[1]] * up).astype(F32), axis=-1, dtype=F32)], axis=1)
        else:
            up = resample(amplitudes[:, i * frame_len:(i + 1) * frame_len], frame_len * upsampling)
            signal = np.concatenate([signal, np.sum((nb * up).astype(F32), axis=-1, dtype=F32)], axis=1)
    return signal[:, :n_samples]


# ---------------------------------------------------------------------------------------------
# Input edge: piano roll -> polyphonic conditioning.
#
# *** This part IS pinned ***: ddsp_piano/utils/midi_encoders.py is plain NumPy, so the reference
# class itself was run in the build container and its inputs/outputs are committed as
# tests/golden/midi_conditioning.npz (generator: tests/golden/make_golden_midi.py).
# ---------------------------------------------------------------------------------------------

class MIDIRoll2Conditioning:
    """Frame-sequential voice allocator, restating ddsp_piano/utils/midi_encoders.py:4-104.

    State (same names as the reference object's attributes): ``assigned_pitch[c]`` = pitch value a
    channel holds (0 = free), ``assigner`` = next free channel (-1 = none), ``reorder`` = last
    channel -> sorted-position permutation."""

    def __init__(self, n_synths=16):
        self.n_synths = int(n_synths)
        self.pitch_mul = np.arange(21, 21 + 88)                    # :19
        self.reorder = list(range(self.n_synths))                  # :20
        self.assigner = 0                                          # :21
        self.assigned_pitch = [0.0] * self.n_synths                # :22

    def update_assigner(self):                                     # :24-32
        n = self.n_synths
        self.assigner = (self.assigner + 1) % n
        if 0.0 not in self.assigned_pitch:
            self.assigner = -1
            return
        while self.assigned_pitch[self.assigner] != 0.0:
            self.assigner = (self.assigner + 1) % n

    def __call__(self, roll):
        roll = np.asarray(roll)
        n = self.n_synths
        activity = roll[..., 0]
        polyphony = np.sum(activity, axis=-1)                      # :47
        scaled = (activity * self.pitch_mul).astype(roll.dtype)    # :49 (in place in the reference)
        # :52-55 -- the n largest values per frame, ascending.  Ties (equal values) are resolved by key
        # number here; the reference leaves them to an unstable argsort (they only matter when equal-valued
        # keys carry different velocities, which a real roll never has: silent keys have velocity 0).
        keys = np.argsort(scaled, axis=-1, kind='stable')[:, 88 - n:]
        top_pitch = np.take_along_axis(scaled, keys, axis=-1)
        top_vel = np.take_along_axis(roll[..., 1], keys, axis=-1)
        out = np.zeros((roll.shape[0], n, 2), dtype=roll.dtype)
        for t in range(roll.shape[0]):
            pitches = [float(v) for v in top_pitch[t]]
            held = self.assigned_pitch
            if t > 0 and set(pitches) == set(held):                # :61-69 nothing started or stopped
                perm = self.reorder
            else:
                perm = [0] * n                                     # :72
                for c in range(n):                                 # :74-79 channels whose note has ended
                    if held[c] not in pitches:
                        held[c] = 0.0
                        if self.assigner == -1:
                            self.update_assigner()
                for c in range(n):                                 # :82-85 sounding notes keep their channel
                    if pitches[c] != 0.0 and pitches[c] in held:
                        perm[held.index(pitches[c])] = c
                for c in range(n):                                 # :88-92 new notes take the next free channel
                    if pitches[c] not in held:
                        perm[self.assigner] = c                    # (-1 addresses the last channel, as in NumPy)
                        held[self.assigner] = pitches[c]
                        self.update_assigner()
                for c in range(n):                                 # :95-98 silence fills the free channels
                    if pitches[c] == 0.0:
                        perm[self.assigner] = c
                        self.update_assigner()
                self.reorder = perm                                # :102
            out[t, :, 0] = top_pitch[t][perm]                      # :66 / :100
            out[t, :, 1] = top_vel[t][perm]                        # :67 / :101
        return out, polyphony


def ensure_sequence_length(sequence, length, right=True):
    """ddsp_piano/utils/io_utils.py:204-224: crop or zero-pad along time, at the end or at the start."""
    sequence = np.asarray(sequence)
    n = sequence.shape[0]
    if n >= length:
        return sequence[:length] if right else sequence[n - length:]
    extra = np.zeros((length - n,) + sequence.shape[1:], dtype=sequence.dtype)
    return np.concatenate([sequence, extra] if right else [extra, sequence], axis=0)
