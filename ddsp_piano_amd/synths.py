"""Synth processors of the reference, same constructor / get_controls / get_signal signatures:

  InHarmonic, MultiInharmonic, MultiAdd      ddsp_piano/modules/inharm_synth.py:130-309
  FilteredNoise                              ddsp.synths.FilteredNoise (default_model.py:44)
  DynamicSizeFilteredNoise                   ddsp_piano/modules/filtered_noise_synth.py:12-42
"""
from __future__ import annotations

import itertools

import torch

from . import _lib, core
from .core import _lib_, _ptr, _stream
from .processors import Processor


class InHarmonic(Processor):
    """inharm_synth.py:130-244."""

    def __init__(self, frame_rate=250, sample_rate=16000, min_frequency=20, scale_fn=core.exp_sigmoid,
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
        return int(self.sample_rate / self.frame_rate)               # :163-165

    def _controls(self, amplitudes, harmonic_distribution, inharm_coef, f0_hz, want_counts=False, want_shifts=True,
                  last_voice_of=None, sparse_for_bank=None):
        """One fused kernel for :183-214 (+ :269); f0_hz may carry several sub-strings.
        want_shifts=False: 'harmonic_shifts' is left out (the compacted oscillator bank forms it from inharm_coef per
        lane and frame, a [R, T, H] tensor less to write and read back); '_inharm_coef' carries the raw coefficients.
        last_voice_of=(n_voices, voice_major) with want_shifts=False: '_shifts_last' [rows / n_voices, T, H] holds the
        harmonic_shifts of every segment's last voice (what the reference's outputs dictionary keeps).
        sparse_for_bank=(n_voices, voice_major) (round 6; with want_counts and want_shifts=False): the caller's only reader
        of 'harmonic_distribution' is the compacted oscillator bank, which takes harmonics at or above a frame's audible
        count as silent from the count: those values are not written (ddspp_inharmonic_controls_sparse) -- except for
        every segment's last voice when last_voice_of is given.  The tensor is then NOT valid for any other reader."""
        amplitudes = core.tf_float32(amplitudes)
        harmonic_distribution = core.tf_float32(harmonic_distribution)
        inharm_coef = core.tf_float32(inharm_coef)
        f0_hz = core.tf_float32(f0_hz)
        if harmonic_distribution.dim() != 3:
            raise ValueError('harmonic_distribution must be [batch, time, n_harmonics]')
        b, t, h = harmonic_distribution.shape
        s = f0_hz.shape[-1]
        for name, x in (('amplitudes', amplitudes), ('inharm_coef', inharm_coef)):
            if tuple(x.shape) != (b, t, 1):
                raise ValueError(f'{name} must have shape {(b, t, 1)}, got {tuple(x.shape)}')
        if f0_hz.dim() != 3 or tuple(f0_hz.shape[:2]) != (b, t):
            raise ValueError(f'f0_hz must be [{b}, {t}, n_substrings], got {tuple(f0_hz.shape)}')
        kind = core.scale_kind(self.scale_fn)
        if kind is None:            # arbitrary python scale_fn: apply it, then let the kernel do the rest
            amplitudes = core.tf_float32(self.scale_fn(amplitudes))
            harmonic_distribution = core.tf_float32(self.scale_fn(harmonic_distribution))
            kind = core.scale_kind(None)
        code, prm = kind
        amp_out = torch.empty_like(amplitudes)
        hd_out = torch.empty_like(harmonic_distribution)
        shifts_out = torch.empty_like(harmonic_distribution) if want_shifts else None
        counts = torch.empty((b, t), dtype=torch.int32, device=amplitudes.device) if want_counts else None
        shifts_last = None
        if (last_voice_of is not None or sparse_for_bank is not None) and not want_shifts:
            n_voices, voice_major = last_voice_of if last_voice_of is not None else sparse_for_bank
            if last_voice_of is not None:
                shifts_last = torch.empty((b // n_voices, t, h), dtype=torch.float32, device=amplitudes.device)
            entry = _lib_().ddspp_inharmonic_controls_sparse if (sparse_for_bank is not None and counts is not None) \
                else _lib_().ddspp_inharmonic_controls_group
            _lib.check(entry(
                _ptr(amplitudes), _ptr(harmonic_distribution), _ptr(inharm_coef), _ptr(f0_hz), _ptr(amp_out),
                _ptr(hd_out), _ptr(shifts_last) if shifts_last is not None else None,
                counts.data_ptr() if counts is not None else None, b, t, h, s,
                int(n_voices), int(bool(voice_major)), float(self.sample_rate), float(self.min_frequency),
                code, prm['exponent'], prm['max_value'], prm['threshold'], prm['gain'],
                int(bool(self.normalize_after_nyquist_cut)), int(bool(self.normalize_below_nyquist)), _stream()))
        else:
            _lib.check(_lib_().ddspp_inharmonic_controls(
                _ptr(amplitudes), _ptr(harmonic_distribution), _ptr(inharm_coef), _ptr(f0_hz), _ptr(amp_out),
                _ptr(hd_out), _ptr(shifts_out), counts.data_ptr() if counts is not None else None, b, t, h, s,
                float(self.sample_rate), float(self.min_frequency),
                code, prm['exponent'], prm['max_value'], prm['threshold'], prm['gain'],
                int(bool(self.normalize_after_nyquist_cut)), int(bool(self.normalize_below_nyquist)), _stream()))
        ctl = {'amplitudes': amp_out, 'harmonic_distribution': hd_out, 'harmonic_shifts': shifts_out,
               'f0_hz': f0_hz}
        if not want_shifts:
            ctl['_inharm_coef'] = inharm_coef
            ctl['_shifts_last'] = shifts_last
        if want_counts:
            ctl['_audible'] = counts       # per-frame count of leading non-silent harmonics (batched route only)
        return ctl

    def get_controls(self, amplitudes, harmonic_distribution, inharm_coef, f0_hz):
        """inharm_synth.py:167-219."""
        f0_hz = core.tf_float32(f0_hz)
        if f0_hz.shape[-1] != 1:
            raise ValueError('InHarmonic expects f0_hz of shape [batch, time, 1]')
        return self._controls(amplitudes, harmonic_distribution, inharm_coef, f0_hz)

    def _synthesize(self, amplitudes, harmonic_distribution, harmonic_shifts, f0_hz):
        amplitudes = core.tf_float32(amplitudes)
        harmonic_distribution = core.tf_float32(harmonic_distribution)
        harmonic_shifts = core.tf_float32(harmonic_shifts)
        f0_hz = core.tf_float32(f0_hz)
        b, t, s = f0_hz.shape
        n_samples = self.upsampling * t                                # :240
        if core.fused_synthesis_supported(t, n_samples) and amplitudes.shape[-1] == 1:
            return core.harmonic_synthesis_fused(f0_hz, amplitudes.reshape(b, t).contiguous(),
                                                 harmonic_distribution, harmonic_shifts, n_samples,
                                                 self.sample_rate, self.inference)
        audio = None
        for sub in range(s):                                           # :279-292
            a = core.harmonic_synthesis(frequencies=f0_hz[..., sub:sub + 1], amplitudes=amplitudes,
                                        harmonic_shifts=harmonic_shifts,
                                        harmonic_distribution=harmonic_distribution, n_samples=n_samples,
                                        sample_rate=self.sample_rate, use_angular_cumsum=self.inference)
            audio = a if audio is None else audio + a
        return audio

    def get_signal(self, amplitudes, harmonic_distribution, harmonic_shifts, f0_hz):
        """inharm_synth.py:221-244."""
        return self._synthesize(amplitudes, harmonic_distribution, harmonic_shifts, f0_hz)


class MultiInharmonic(InHarmonic):
    """Inharmonic synthesizer with multiple F0 controls -- inharm_synth.py:247-293."""

    def __init__(self, name='multi_inharmonic', **kwargs):
        super().__init__(name=name, **kwargs)

    def get_controls(self, amplitudes, harmonic_distribution, inharm_coef, f0_hz):
        return self._controls(amplitudes, harmonic_distribution, inharm_coef, f0_hz)    # :254-270

    def get_signal(self, amplitudes, harmonic_distribution, harmonic_shifts, f0_hz):
        return self._synthesize(amplitudes, harmonic_distribution, harmonic_shifts, f0_hz)  # :272-293


class SurrogateAdditive(Processor):
    """ddsp_piano/modules/surrogate_synth.py:107-214 (only configs/surrogate.gin uses it): the inharmonic
    bank with per-harmonic decaying amplitudes.  get_controls is composed from the library's scale /
    mask primitives, get_signal runs resample -> decay envelope -> cos_oscillator_bank kernels."""

    def __init__(self, frame_rate=250, sample_rate=16000, min_frequency=20, normalize_harm_distribution=True,
                 scale_fn=core.exp_sigmoid, normalize_below_nyquist=True, inference=False, name='inharmonic'):
        super().__init__(name=name)
        self.frame_rate, self.sample_rate, self.min_frequency = frame_rate, sample_rate, min_frequency
        self.normalize_harm_distribution = normalize_harm_distribution
        self.scale_fn = scale_fn
        self.normalize_below_nyquist = normalize_below_nyquist
        self.inference = inference

    def get_controls(self, amplitudes, decays, decay_time, harmonic_distribution, inharm_coef, f0_hz, _want_counts=False):
        """surrogate_synth.py:140-197.  (_want_counts, batched route only: '_audible' = the per-frame counts of leading
        non-silent harmonics and '_inharm_coef' = the raw coefficients, as InHarmonic._controls leaves them for the
        compacted bank; None when the kernels did not run.)  Two kernels when the shapes are the plain ones: ddspp_inharmonic_controls (scale,
        shifts, Nyquist cut, audibility gate, normalisation -- mode 2 = none for normalize_harm_distribution=False) and
        ddspp_surrogate_decays; anything else is composed from the library's primitives as before."""
        amplitudes, harmonic_distribution = core.tf_float32(amplitudes), core.tf_float32(harmonic_distribution)
        f0_hz = core.tf_float32(f0_hz)
        kind = core.scale_kind(self.scale_fn)
        if (kind is not None and harmonic_distribution.dim() == 3 and harmonic_distribution.is_cuda
                and harmonic_distribution.shape[-1] <= 512
                and tuple(amplitudes.shape) == tuple(harmonic_distribution.shape[:2]) + (1,)
                and tuple(f0_hz.shape) == tuple(amplitudes.shape)
                and inharm_coef is not None and tuple(inharm_coef.shape) == tuple(amplitudes.shape)
                and (decays is None or tuple(decays.shape) == tuple(harmonic_distribution.shape))):
            b, t, h = harmonic_distribution.shape
            code, prm = kind
            amplitudes, harmonic_distribution = amplitudes.contiguous(), harmonic_distribution.contiguous()
            inharm_coef, f0_hz = core.tf_float32(inharm_coef).contiguous(), f0_hz.contiguous()
            amp_out, hd_out = torch.empty_like(amplitudes), torch.empty_like(harmonic_distribution)
            shifts_out = torch.empty_like(harmonic_distribution)
            counts = torch.empty((b, t), dtype=torch.int32, device=amplitudes.device) if _want_counts else None
            _lib.check(_lib_().ddspp_inharmonic_controls(
                _ptr(amplitudes), _ptr(harmonic_distribution), _ptr(inharm_coef), _ptr(f0_hz), _ptr(amp_out), _ptr(hd_out),
                _ptr(shifts_out), counts.data_ptr() if counts is not None else None, b, t, h, 1, float(self.sample_rate),
                float(self.min_frequency), code,
                prm['exponent'], prm['max_value'], prm['threshold'], prm['gain'],
                1 if self.normalize_harm_distribution else 2, int(bool(self.normalize_below_nyquist)), _stream()))
            if decays is not None:
                decays = core.tf_float32(decays).contiguous()
                dec_out = torch.empty_like(decays)
                _lib.check(_lib_().ddspp_surrogate_decays(_ptr(decays), _ptr(inharm_coef), _ptr(f0_hz), _ptr(dec_out), b, t, h,
                                                          float(self.sample_rate), _stream()))
                decays = dec_out
            ctl = {'amplitudes': amp_out, 'decays': decays, 'decay_time': decay_time,
                   'harmonic_distribution': hd_out, 'harmonic_shifts': shifts_out, 'f0_hz': f0_hz}
            if _want_counts:
                ctl['_audible'], ctl['_inharm_coef'] = counts, inharm_coef
            return ctl
        if self.scale_fn is not None:                                                   # :152-154
            amplitudes = core.tf_float32(self.scale_fn(amplitudes))
            harmonic_distribution = core.tf_float32(self.scale_fn(harmonic_distribution))
        inharm_coef = torch.clamp(core.tf_float32(inharm_coef), min=0.0)                 # :157
        n_harmonics = int(harmonic_distribution.shape[-1])
        inharmonic_freq, harmonic_shifts = core.get_inharmonic_freq(f0_hz, inharm_coef, n_harmonics)
        if decays is not None:                                                          # :163-171
            decays = torch.clamp(core.tf_float32(decays), min=1e-5, max=1.0)
            decays = torch.where(inharmonic_freq >= self.sample_rate / 2.0, torch.ones_like(decays), decays)
        if self.normalize_below_nyquist:                                                # :172-181
            harmonic_distribution = core.remove_above_nyquist(inharmonic_freq, harmonic_distribution,
                                                              self.sample_rate)
            amplitudes = amplitudes * (f0_hz > self.min_frequency).to(torch.float32)
        if self.normalize_harm_distribution:                                            # :183-187
            harmonic_distribution = core.safe_divide(harmonic_distribution,
                                                     harmonic_distribution.sum(dim=-1, keepdim=True))
        return {'amplitudes': amplitudes, 'decays': decays, 'decay_time': decay_time,
                'harmonic_distribution': harmonic_distribution, 'harmonic_shifts': harmonic_shifts, 'f0_hz': f0_hz}

    def get_signal(self, amplitudes, decays, decay_time, harmonic_distribution, harmonic_shifts, f0_hz):
        return core.surrogate_harmonic_synthesis(
            frequencies=f0_hz, amplitudes=amplitudes, decays=decays, decay_time=decay_time,
            harmonic_shifts=harmonic_shifts, harmonic_distribution=harmonic_distribution,
            upsampling=int(self.sample_rate / self.frame_rate), sample_rate=self.sample_rate,
            use_angular_cumsum=self.inference)


class MultiAdd(Processor):
    """Sum arbitrary number of signals -- inharm_synth.py:296-309."""

    def __init__(self, name='add'):
        super().__init__(name=name)

    def get_controls(self, *signals):
        return {f'signal_{i}': s for i, s in enumerate(signals)}

    def get_signal(self, **signals):
        return core.add_signals(list(signals.values()))


class FilteredNoise(Processor):
    """ddsp.synths.FilteredNoise(n_samples, window_size, scale_fn, initial_bias, name).

    The reference draws an unseeded ``tf.random.uniform([B, n_samples], -1, 1)`` per call.  Here the
    draw comes from the library's Philox generator keyed by ``seed`` with a per-call counter, or from
    the explicit ``noise=`` argument of get_signal (what the parity tests use).
    """

    def __init__(self, n_samples=64000, window_size=257, scale_fn=core.exp_sigmoid, initial_bias=-5.0,
                 name='filtered_noise', seed=0):
        super().__init__(name=name)
        self.n_samples = n_samples
        self.window_size = window_size
        self.scale_fn = scale_fn
        self.initial_bias = initial_bias
        self.seed = seed
        self._calls = itertools.count()

    def raw_scale(self):
        """(kind, bias, params) when scale_fn is one of the library's (so get_controls can be fused into the FIR
        design kernel), else None."""
        if self.scale_fn is None:
            return None
        kind = core.scale_kind(self.scale_fn)
        return None if kind is None else (kind[0], float(self.initial_bias), kind[1])

    def get_controls(self, magnitudes):
        magnitudes = core.tf_float32(magnitudes)
        if self.scale_fn is not None:
            rs = self.raw_scale()
            if rs is None:
                magnitudes = core.tf_float32(self.scale_fn(magnitudes + self.initial_bias))
            else:
                magnitudes = core.scale_bias(magnitudes, *rs)
        return {'magnitudes': magnitudes}

    def _n_samples(self, magnitudes):
        return int(self.n_samples)

    def draw_noise(self, batch_size, n_samples, device, out=None):
        call = next(self._calls)
        return core.uniform_noise((batch_size, n_samples), seed=self.seed, offset=call << 40, device=device, out=out)

    def draw_noise_lazy(self, batch_size, n_samples, device):
        """The same draw as draw_noise (same call counter, same numbers), left to the filter kernel (core.DrawnNoise)."""
        call = next(self._calls)
        return core.DrawnNoise(batch_size, n_samples, self.seed, call << 40, device)

    def get_signal(self, magnitudes, noise=None):
        magnitudes = core.tf_float32(magnitudes)
        batch_size = int(magnitudes.shape[0])
        n_samples = self._n_samples(magnitudes)
        if noise is None:
            noise = self.draw_noise_lazy(batch_size, n_samples, magnitudes.device)
        else:
            noise = core.tf_float32(noise)
            if tuple(noise.shape) != (batch_size, n_samples):
                raise ValueError(f'noise must be {(batch_size, n_samples)}, got {tuple(noise.shape)}')
        return core.frequency_filter(noise, magnitudes, window_size=self.window_size)


class DynamicSizeFilteredNoise(FilteredNoise):
    """filtered_noise_synth.py:12-42: n_samples = upsampling * n_frames."""

    def __init__(self, frame_rate=250, sample_rate=16000, **kwargs):
        kwargs.setdefault('name', 'filtered_noise')
        super().__init__(**kwargs)
        self.frame_rate = frame_rate
        self.sample_rate = sample_rate

    @property
    def upsampling(self):
        return int(self.sample_rate / self.frame_rate)

    def _n_samples(self, magnitudes):
        return self.upsampling * int(magnitudes.shape[1])             # :35-37
