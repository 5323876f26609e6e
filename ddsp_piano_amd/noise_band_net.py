"""NoiseBandNetSynth and its FilterBank -- ddsp_piano/modules/filtered_noise_synth.py:51-317 (a tf-ddsp adaptation of
adrianbarahona/noisebandnet).  Not referenced by any shipped gin file; SURVEY.md 8f-3.

Set-up (filter design, the loopable noise bands) is host work done once per (n_band, sample_rate, ...), with the same
scipy.signal calls the reference makes (kaiserord, firwin).  The per-call work -- chunk-wise linear upsampling of the
band amplitudes, modulation of the noise bands, sum over bands -- is one HIP kernel (csrc/noise_bands.hip).

Two things the reference leaves to TensorFlow's random generator are explicit here (as for FilteredNoise): the phases of
the noise bands (tf.random.uniform(seed=42), :289-291 -- a seeded TF stream cannot be reproduced outside TF) come from
``phase_noise=`` or numpy's Philox(42); the per-call roll of the bands (:224-231, unseeded) comes from ``shift=`` of
get_signal or the synth's own counter-based draw."""
from __future__ import annotations

import itertools
import math

import numpy as np
import torch

from . import _lib, core
from .core import _lib_, _ptr, _stream
from .processors import Processor


class FilterBank:
    """filtered_noise_synth.py:51-158: linearly + logarithmically spaced Kaiser-windowed FIR band filters."""

    def __init__(self, n_filters_linear=1024, n_filters_log=1024, linear_min_f=20, linear_max_f_cutoff_fs=4,
                 attenuation=50, sample_rate=16000):
        bands = self.get_frequency_bands(n_filters_linear, n_filters_log, linear_min_f, linear_max_f_cutoff_fs, sample_rate)
        self.band_centers = self.get_band_centers(bands, sample_rate)
        self.filters = self.build_filterbank(bands, sample_rate, attenuation)
        self.max_filter_len = max(len(h) for h in self.filters)

    @staticmethod
    def get_frequency_bands(n_filters_linear, n_filters_log, linear_min_f, linear_max_f_cutoff_fs, sample_rate):
        linear_max_f = (sample_rate / 2) / linear_max_f_cutoff_fs                     # :88-92
        lin = np.linspace(linear_min_f, linear_max_f, n_filters_linear)
        lin = np.vstack((lin[:-1], lin[1:])).T
        if linear_max_f_cutoff_fs == 1:
            # the reference returns an undefined name here (`linear_center_f`, :108-109) and dies with a NameError;
            # a purely linear bank was clearly meant, so that is what this returns
            return lin
        log = np.geomspace(start=linear_max_f, stop=sample_rate / 2, num=n_filters_log, endpoint=False)   # :94-101
        log = np.vstack((log[:-1], log[1:])).T
        return np.concatenate((lin, log))

    @staticmethod
    def get_band_centers(frequency_bands, sample_rate):                                # :115-119
        mean = np.mean(frequency_bands, axis=1)
        return np.concatenate(([frequency_bands[0, 0] / 2], mean, [((sample_rate / 2) + frequency_bands[-1, -1]) / 2]))

    @staticmethod
    def get_filter(cutoff, sample_rate, attenuation, pass_zero, transition_bandwidth=0.2, scale=True):   # :121-133
        from scipy import signal
        if isinstance(cutoff, np.ndarray):
            bandwidth = abs(cutoff[1] - cutoff[0])
        elif pass_zero:
            bandwidth = cutoff
        else:
            bandwidth = abs((sample_rate / 2) - cutoff)
        width = (bandwidth / (sample_rate / 2)) * transition_bandwidth
        n, beta = signal.kaiserord(ripple=attenuation, width=width)
        n = 2 * (n // 2) + 1
        return signal.firwin(numtaps=n, cutoff=cutoff, window=('kaiser', beta), scale=scale, fs=sample_rate,
                             pass_zero=pass_zero)

    def build_filterbank(self, frequency_bands, sample_rate, attenuation):             # :135-158
        filters = []
        last = frequency_bands.shape[0] - 1
        for i in range(frequency_bands.shape[0]):
            if i == 0:
                filters.append(self.get_filter(frequency_bands[i, 0], sample_rate, attenuation, pass_zero=True))
            filters.append(self.get_filter(frequency_bands[i], sample_rate, attenuation, pass_zero=False))
            if i == last:
                filters.append(self.get_filter(frequency_bands[i, -1], sample_rate, attenuation, pass_zero=False))
        return filters


def get_next_power_of_2(x):
    return int(math.pow(2, math.ceil(math.log(x) / math.log(2))))


def get_noise_bands(fb, min_noise_len, normalize, phase_noise=None, seed=42):
    """filtered_noise_synth.py:283-309: deterministic loopable noise bands [noise_len, n_band] (float32) and noise_len.
    phase_noise [n_band, noise_len // 2 + 1] in (-pi, pi): explicit phases; None: numpy Philox(seed)."""
    noise_len = get_next_power_of_2(fb.max_filter_len) if fb.max_filter_len > min_noise_len else min_noise_len
    filters = np.stack([np.pad(h, (noise_len - len(h), 0)) for h in fb.filters]).astype(np.float32)   # pad_filters
    mag = np.abs(np.fft.rfft(filters.astype(np.float64), axis=-1)).astype(np.float32)
    if phase_noise is None:
        rng = np.random.Generator(np.random.Philox(seed))
        phase_noise = rng.uniform(-math.pi, math.pi, size=mag.shape).astype(np.float32)
    phase_noise = np.asarray(phase_noise, np.float32)
    if phase_noise.shape != mag.shape:
        raise ValueError(f'phase_noise must be {mag.shape}, got {phase_noise.shape}')
    ph = np.exp(1j * phase_noise.astype(np.float64))
    ph[:, 0] = 0.0
    ph[:, -1] = 0.0
    bands = np.fft.irfft(mag.astype(np.float64) * ph, n=noise_len, axis=-1).astype(np.float32)
    if normalize:
        bands = (bands / np.max(np.abs(bands))).astype(np.float32)
    return np.ascontiguousarray(bands.T), noise_len


class NoiseBandNetSynth(Processor):
    """filtered_noise_synth.py:161-262."""

    def __init__(self, upsampling=64, filterbank_attenuation=50, sample_rate=16000, min_noise_len=2 ** 4, linear_min_f=20,
                 linear_max_f_cutoff_fs=4, normalize_noise_bands=True, scale_fn=core.exp_sigmoid, inference=False,
                 name='noise', phase_noise=None, seed=42):
        super().__init__(name=name)
        if not (isinstance(min_noise_len, int) and min_noise_len > 0 and 2 ** int(math.log(min_noise_len, 2)) == min_noise_len):
            raise AssertionError('min_noise_len must be a positive integer and a power of 2')
        self.scale_fn = scale_fn
        self.upsampling = upsampling
        self.sample_rate = sample_rate
        self.linear_min_f = linear_min_f
        self.linear_max_f_cutoff_fs = linear_max_f_cutoff_fs
        self.filterbank_attenuation = filterbank_attenuation
        self.min_noise_len = min_noise_len
        self.normalize_noise_bands = normalize_noise_bands
        self.inference = inference
        self.phase_noise = phase_noise
        self.seed = seed
        self.n_band = None
        self.noise_bands = None                         # [noise_len, n_band] device tensor once built
        self._calls = itertools.count()
        self._tables = {}

    def build(self, n_band, device=None):
        """:194-211 -- the bank for n_band amplitude channels."""
        self.n_band = int(n_band)
        fb = FilterBank(n_filters_linear=self.n_band // 2, n_filters_log=self.n_band // 2, linear_min_f=self.linear_min_f,
                        linear_max_f_cutoff_fs=self.linear_max_f_cutoff_fs, sample_rate=self.sample_rate,
                        attenuation=self.filterbank_attenuation)
        self.center_frequencies = fb.band_centers
        bands, self.noise_len = get_noise_bands(fb, self.min_noise_len, self.normalize_noise_bands, self.phase_noise, self.seed)
        if bands.shape[1] != self.n_band:
            raise ValueError(f'the filter bank has {bands.shape[1]} bands for {self.n_band} amplitude channels '
                             '(n_band must be even and >= 4)')
        self.noise_bands = torch.from_numpy(bands).to(device or core.default_device())

    def get_controls(self, magnitudes):
        magnitudes = core.tf_float32(magnitudes)
        if self.scale_fn is not None:
            magnitudes = core.tf_float32(self.scale_fn(magnitudes))
        return {'amplitudes': magnitudes}

    def _resample_tables(self, n_frames, device):
        """Per-sample source frames / weights of the reference's chunk-wise core.resample calls (:233-259)."""
        key = (n_frames, self.noise_len, self.upsampling, core.RECALLED['resize'], str(device))
        tab = self._tables.get(key)
        if tab is None:
            frame_len = int(self.noise_len / self.upsampling)
            n_samples = n_frames * self.upsampling
            rule = core.RECALLED['resize']
            los, his, ws = [], [], []
            if n_frames / frame_len < 1:
                lo, hi, w, _ = core._linear_tables_np(n_frames, n_samples, rule)
                los, his, ws = [lo], [hi], [w]
            else:
                for i in range(math.ceil(n_frames / frame_len)):
                    t_c = min(frame_len, n_frames - i * frame_len)          # a short last chunk is stretched to a full one
                    lo, hi, w, _ = core._linear_tables_np(t_c, frame_len * self.upsampling, rule)
                    los.append(lo + i * frame_len)
                    his.append(hi + i * frame_len)
                    ws.append(w)
            lo = np.concatenate(los)[:n_samples].astype(np.int32)
            hi = np.concatenate(his)[:n_samples].astype(np.int32)
            w = np.concatenate(ws)[:n_samples].astype(np.float32)
            tab = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in (lo, hi, w))
            self._tables[key] = tab
        return tab

    def get_signal(self, amplitudes, shift=None):
        amplitudes = core.tf_float32(amplitudes)
        if amplitudes.dim() != 3:
            raise ValueError('amplitudes must be [batch, n_frames, n_band]')
        b, t, k = amplitudes.shape
        if self.noise_bands is None or self.n_band != k or self.noise_bands.device != amplitudes.device:
            self.build(k, amplitudes.device)
        if shift is None:                     # the reference draws it unseeded per call (:226-230)
            call = next(self._calls)
            shift = int(np.random.Generator(np.random.Philox(key=self.seed + 1, counter=call)).integers(0, self.noise_len))
        n_samples = t * self.upsampling
        lo, hi, w = self._resample_tables(t, amplitudes.device)
        out = torch.empty((b, n_samples), dtype=torch.float32, device=amplitudes.device)
        _lib.check(_lib_().ddspp_noise_bands(_ptr(amplitudes), _ptr(self.noise_bands), _ptr(lo), _ptr(hi),
                                             _ptr(w), _ptr(out), b, t, k, n_samples, int(self.noise_len),
                                             int(shift) % int(self.noise_len), _stream()))
        return out
