"""Host side of the hot path: the ``ddsp.core`` functions the reference calls, same names and
argument meaning, evaluated by the HIP kernels of libddspp on torch-ROCm buffers.

Reference call sites (all under /root/reference/ddsp_piano/modules):
  inharm_synth.py:20-46   get_inharmonic_freq
  inharm_synth.py:49-84   cos_oscillator_bank            (core.remove_above_nyquist, core.angular_cumsum)
  inharm_synth.py:87-127  harmonic_synthesis             (core.get_harmonic_frequencies, core.resample)
  filtered_noise_synth.py:41-42  core.frequency_filter   (frequency_impulse_response, fft_convolve)
  fdn_reverb.py:409       fft_convolve
Everything here is float32; Python only validates shapes (raising what ddsp raises), builds the
small index / window tables with the float32 arithmetic of the TF kernels, and enqueues kernels on
the current HIP stream.
"""
from __future__ import annotations

import atexit
import ctypes
import functools
import math
import threading

import numpy as np
import torch

from . import _lib

F32 = np.float32
_TWO_PI_F32 = F32(2.0 * np.pi)


# ----------------------------------------------------------------------------------------------------
# Details of the un-vendored ddsp 3.7.0 that are restated from memory (SURVEY.md 8(c) "VERIFY" list; ddsp is not
# installable next to this package).  Where the LIBRARY decides, the rule is a switch here, mirrored one to one by
# oracle.ddsp_oracle.RECALLED; the defaults are what this package believes ddsp 3.7.0 does.
#   auto_delay   crop_and_compensate_delay with delay_compensation < 0: 'ddsp370' (ir_size - 1) // 2 - 1 | 'half'
#                ir_size // 2.  C-ABI: delay_compensation = DDSPP_DELAY_AUTO (-1) / DDSPP_DELAY_AUTO_HALF (-2).
#   resize       core.resample(method='linear'): 'legacy' (TF1 bilinear, pos = n * T/N) | 'half_pixel'.
#                Host tables; the fused oscillator path needs 'legacy' (else the three-operator route runs).
#   window_crop  apply_window_to_impulse_response, window_size < ir_size: 'ddsp370' | 'centred'.  Host matrix.
#   angular_offsets  core.angular_cumsum's running sum of chunk end phases: 'wrapped' (`tf.cumsum(offsets, axis=1) %
#                (2 pi)`) | 'plain' (the sum is added to the chunk as it is).  Library option
#                DDSPP_ANGULAR_OFFSETS_PLAIN (ddspp_set_option; every oscillator kernel reads it at launch, round 5).
# One more switch of the same kind is not about ddsp but about TensorFlow's arithmetic:
#   fdn_solve    FeedbackDelayNetwork.get_late_ir's per-bin 8 x 8 system: 'float64' (solved in double: the value the
#                reference's recipe approximates) | 'complex64' (tf.linalg.inv + matmuls in complex64 as
#                fdn_reverb.py:314-333 writes them).  C-ABI: ddspp_fdn_transfer(solve).
# The other three recalled items are ordinary arguments here (exp_sigmoid's exponent / max_value / threshold,
# FilteredNoise(initial_bias=)) or oracle-only (the inclusive scan of angular_cumsum is what the kernels implement).
# ----------------------------------------------------------------------------------------------------
RECALLED = {'auto_delay': 'ddsp370', 'resize': 'legacy', 'window_crop': 'ddsp370', 'fdn_solve': 'float64',
            'angular_offsets': 'wrapped'}
_RECALLED_CHOICES = {'auto_delay': ('ddsp370', 'half'), 'resize': ('legacy', 'half_pixel'),
                     'window_crop': ('ddsp370', 'centred'), 'fdn_solve': ('float64', 'complex64'),
                     'angular_offsets': ('wrapped', 'plain')}


def set_recalled(**rules):
    """Select another recollection of a ddsp detail (see RECALLED); returns the previous settings."""
    for k, v in rules.items():
        if k not in _RECALLED_CHOICES:
            raise KeyError(f'unknown recalled detail {k!r}; known: {sorted(_RECALLED_CHOICES)}')
        if v not in _RECALLED_CHOICES[k]:
            raise ValueError(f'{k} must be one of {_RECALLED_CHOICES[k]}, got {v!r}')
    previous = dict(RECALLED)
    # the library's switch first, and only when it changes (a restore of `previous` on a host-only path must not load
    # libddspp): RECALLED says 'plain' only once the kernels do
    if 'angular_offsets' in rules and rules['angular_offsets'] != RECALLED['angular_offsets']:
        _lib.set_option('DDSPP_ANGULAR_OFFSETS_PLAIN', 1 if rules['angular_offsets'] == 'plain' else 0, persistent=True)
    RECALLED.update(rules)
    return previous


def _auto_delay(delay_compensation):
    """The C-ABI code of a delay_compensation argument: >= 0 as given, < 0 -> the selected automatic rule."""
    if delay_compensation >= 0:
        return int(delay_compensation)
    return -2 if RECALLED['auto_delay'] == 'half' else -1


# ----------------------------------------------------------------------------------------------------
# buffer plumbing
# ----------------------------------------------------------------------------------------------------
def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError('ddsp_piano_amd needs an AMD GPU (torch.cuda.is_available() is False); '
                           'there is no CPU fallback for the synthesis path.')
    return torch.device('cuda', torch.cuda.current_device())


def tf_float32(x, device=None):
    """ddsp.core.tf_float32: cast to float32 (and make the buffer a contiguous device tensor)."""
    if isinstance(x, torch.Tensor):
        if x.dtype == torch.float32 and x.is_cuda and x.is_contiguous():
            return x                       # (the common case, a few hundred times per streamed push: no dispatcher round trip)
        if x.is_cuda or (device is None and not torch.cuda.is_available()):
            # already resident (or: no GPU at all -- host-logic tests; kernels refuse CPU buffers)
            return x.to(dtype=torch.float32).contiguous()
        return x.to(device=device or default_device(), dtype=torch.float32).contiguous()
    return torch.as_tensor(np.asarray(x, dtype=np.float32), device=device or default_device()).contiguous()


def _ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError('libddspp kernels take device buffers only: got a CPU tensor '
                           '(there is no CPU fallback for the synthesis path)')
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """The current HIP stream of the current device as the C-ABI takes it (the raw handle: a dozen calls per group call,
    and torch.cuda.current_stream() builds a Stream object each time)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _lib_():
    return _lib.load()


# ----------------------------------------------------------------------------------------------------
# scale functions (usable as `scale_fn=`; the kernels recognise them through `_ddspp_kind`)
# ----------------------------------------------------------------------------------------------------
def exp_sigmoid(x, exponent=10.0, max_value=2.0, threshold=1e-7):
    """ddsp.core.exp_sigmoid: ``max_value * sigmoid(x) ** log(exponent) + threshold``."""
    x = tf_float32(x)
    y = torch.empty_like(x)
    _lib.check(_lib_().ddspp_scale_bias(_ptr(x), _ptr(y), x.numel(), 0.0, 1, exponent, max_value, threshold,
                                        1.0, _stream()))
    return y


exp_sigmoid._ddspp_kind = (1, dict(exponent=10.0, max_value=2.0, threshold=1e-7, gain=1.0))


def exp_tanh(x, max_value=2.0, exponent=10.0, gain=1.0, threshold=1e-7):
    """ddsp_piano/modules/inharm_synth.py:13-17."""
    x = tf_float32(x)
    y = torch.empty_like(x)
    _lib.check(_lib_().ddspp_scale_bias(_ptr(x), _ptr(y), x.numel(), 0.0, 2, exponent, max_value, threshold,
                                        gain, _stream()))
    return y


exp_tanh._ddspp_kind = (2, dict(exponent=10.0, max_value=2.0, threshold=1e-7, gain=1.0))


def scale_kind(scale_fn):
    """(kind, params) for the fused control kernels, or None for an arbitrary python callable."""
    if scale_fn is None:
        return 0, dict(exponent=10.0, max_value=2.0, threshold=1e-7, gain=1.0)
    fn = scale_fn
    kw = {}
    if isinstance(fn, functools.partial):
        if fn.args:
            return None
        kw = dict(fn.keywords)
        fn = fn.func
    kind = getattr(fn, '_ddspp_kind', None)
    if kind is None:
        return None
    params = dict(kind[1])
    for k, v in kw.items():
        if k not in params:
            return None
        params[k] = float(v)
    return kind[0], params


def safe_divide(numerator, denominator, eps=1e-7):
    """ddsp.core.safe_divide."""
    numerator, denominator = tf_float32(numerator), tf_float32(denominator)
    safe = torch.where(denominator == 0.0, torch.full_like(denominator, eps), denominator)
    return numerator / safe


def remove_above_nyquist(frequency_envelopes, amplitude_envelopes, sample_rate=16000):
    """ddsp.core.remove_above_nyquist (``>=``)."""
    f, a = tf_float32(frequency_envelopes), tf_float32(amplitude_envelopes)
    return torch.where(f >= sample_rate / 2.0, torch.zeros_like(a), a)


def get_harmonic_frequencies(frequencies, n_harmonics):
    """ddsp.core.get_harmonic_frequencies: ``f0 * linspace(1, H, H)``."""
    frequencies = tf_float32(frequencies)
    ratios = torch.linspace(1.0, float(n_harmonics), int(n_harmonics), device=frequencies.device,
                            dtype=torch.float32)
    return frequencies * ratios[None, None, :]


def get_inharmonic_freq(f0_hz, inharm_coef, n_harmonics):
    """ddsp_piano/modules/inharm_synth.py:20-46 (each op separately rounded, as in the reference)."""
    f0_hz, inharm_coef = tf_float32(f0_hz), tf_float32(inharm_coef)
    k = torch.linspace(1.0, float(n_harmonics), int(n_harmonics), device=f0_hz.device,
                       dtype=torch.float32)[None, None, :]
    inharm_factor = k * k                                  # tf.math.pow(int_multiplier, 2)
    inharm_factor = inharm_factor * inharm_coef + 1.0
    inharm_factor = torch.sqrt(inharm_factor)
    inharmonic_freq = f0_hz * k * inharm_factor
    harmonic_shifts = inharm_factor - 1.0
    return inharmonic_freq, harmonic_shifts


def midi_to_hz(notes):
    """ddsp.core.midi_to_hz."""
    return 440.0 * (2.0 ** ((np.asarray(notes, dtype=np.float64) - 69.0) / 12.0))


# ----------------------------------------------------------------------------------------------------
# host-built tables (float32 arithmetic of the TF kernels)
# ----------------------------------------------------------------------------------------------------
def _hann_window_np(n, periodic=True):
    """tf.signal.hann_window in float32: ``0.5 - 0.5 * cos(2*pi*i / (n + periodic*even - 1))``."""
    n = int(n)
    if n == 1:
        return np.ones([1], F32)
    even = 1 - n % 2
    denom = F32(n + int(periodic) * even - 1)
    count = np.arange(n, dtype=F32)
    cos_arg = ((_TWO_PI_F32 * count).astype(F32) / denom).astype(F32)
    return (F32(0.5) - (F32(0.5) * np.cos(cos_arg, dtype=F32)).astype(F32)).astype(F32)


@functools.lru_cache(maxsize=64)
def _linear_tables_np(n_frames, n_timesteps, rule='legacy'):
    """Bilinear source rows and weights of tf.compat.v1.image.resize(align_corners=False): the TF1 kernel has no
    half-pixel centres ('legacy', pos = n * T/N); 'half_pixel' is the TF2 scaler, (n + 0.5) * T/N - 0.5."""
    scale = F32(n_frames) / F32(n_timesteps)
    if rule == 'half_pixel':
        pos = (((np.arange(n_timesteps, dtype=F32) + F32(0.5)).astype(F32) * scale).astype(F32) - F32(0.5)).astype(F32)
    else:
        pos = (np.arange(n_timesteps, dtype=F32) * scale).astype(F32)
    fl = np.floor(pos)
    # (TF clamps `lower` only from below: where float32(N - 1) * scale rounds up to T -- the last sample of a file of more
    # than 131 072 frames at hop 96 -- its kernel reads a row past the tensor; that one sample takes the last row here)
    lo = np.minimum(np.maximum(fl.astype(np.int32), 0), n_frames - 1).astype(np.int32)
    hi = np.minimum(np.maximum(np.ceil(pos).astype(np.int32), 0), n_frames - 1).astype(np.int32)
    w = (pos - fl).astype(F32)
    aligned = False
    if n_timesteps % n_frames == 0:
        u = n_timesteps // n_frames
        aligned = bool(np.array_equal(lo, (np.arange(n_timesteps) // u).astype(np.int32)))
    return lo, hi, w, aligned


WALK_BLOCK = 8                  # csrc/osc_common.h BLK: samples per block of the frame-walking kernels
WALK_NEXT_ROW = F32(1.0)        # csrc/osc_common.h: the mark of a sample that takes row t + 1 itself


def _walk_weights_np(n_frames, n_timesteps, rule='legacy', first_sample=0, n=None):
    """(w, walkable): `wlin` as the frame-walking kernels take it -- ddspp_walk_weights_host (csrc/tables.cpp) in numpy.

    The fused / compacted oscillator kernels walk frame t = n // U and interpolate between rows t and t + 1: the resize
    kernel's (lo, hi) pair while floor(float32(n) * float32(T / N)) == n // U.  Far into a long file (frame 131 073 at
    hop 96 -- synthesize_midi_file.py:41-73 on a piece of more than 8.7 minutes) the product rounds up to the next whole
    frame for the last sample(s) of a frame: rows (t + 1, t + 1), weight 0, i.e. x[t + 1] itself; those samples carry
    the mark 1.0 and the kernels substitute x1 exactly.  walkable = False when some sample needs anything else."""
    n_frames, n_timesteps, first_sample = int(n_frames), int(n_timesteps), int(first_sample)
    n = n_timesteps if n is None else int(n)
    scale = F32(n_frames) / F32(n_timesteps)
    idx = np.arange(first_sample, first_sample + n, dtype=np.int64)
    x = idx.astype(F32)
    if rule == 'half_pixel':
        pos = (((x + F32(0.5)).astype(F32) * scale).astype(F32) - F32(0.5)).astype(F32)
    else:
        pos = (x * scale).astype(F32)
    fl = np.floor(pos)
    w = (pos - fl).astype(F32)
    if n_timesteps % n_frames != 0:
        return w, False
    u = n_timesteps // n_frames
    t = idx // u
    lo = np.maximum(fl.astype(np.int64), 0)
    off = lo != t
    if not off.any():
        return w, True
    nxt = off & (lo == t + 1) & (w == 0) & (idx % u >= u - WALK_BLOCK)
    if not np.array_equal(nxt, off):
        return w, False
    w = w.copy()
    w[nxt] = WALK_NEXT_ROW
    return w, True


@functools.lru_cache(maxsize=2)        # (whole-signal float32 arrays: 115 MB for a 20-minute file; the device copy lives in _table_cache)
def _walk_full_np(n_frames, n_timesteps, rule):
    """(w, walkable) of a whole signal, computed ONCE per shape: fused_synthesis_supported() and walk_weights() both ask
    (a 20-minute file is 28.8 M samples -- seconds of host work and ~1 GB of temporaries per evaluation)."""
    return _walk_weights_np(n_frames, n_timesteps, rule)


@functools.lru_cache(maxsize=64)
def _walkable_np(n_frames, n_timesteps, rule):
    if rule == 'legacy' and n_timesteps % n_frames == 0 and n_timesteps < (1 << 24):
        # below 2^24 samples float32(T) / float32(N) is RN(1 / U) exactly, and linear_exact_frames(U) is the first frame at
        # which float32(n) * RN(1 / U) leaves frame n // U: shorter signals are frame aligned, nothing to build
        if n_frames < linear_exact_frames(n_timesteps // n_frames):
            return True
    return _walk_full_np(n_frames, n_timesteps, rule)[1]


@functools.lru_cache(maxsize=8)
def _walkable_piece_np(n_frames, n_timesteps, rule, first_sample, n):
    """walkable flag of samples first_sample .. + n (host arithmetic, a few thousand samples for a streamed piece).  Cached:
    StreamingSynthesizer.push asks core.walkable() and the kernels' wrappers ask again through walk_weights() with the same
    arguments -- the second question costs nothing, and neither touches the device (a device-side check needed a
    synchronising .item() per piece, which is also illegal inside a stream capture: ADVICE r05)."""
    return bool(_walk_weights_np(n_frames, n_timesteps, rule, first_sample, n)[1])


_table_cache = {}
_table_lock = threading.Lock()


def _cached(key, builder):
    with _table_lock:
        val = _table_cache.get(key)
    if val is None:
        val = builder()
        with _table_lock:
            _table_cache[key] = val
    return val


def linear_tables(n_frames, n_timesteps, device):
    rule = RECALLED['resize']

    def build():
        lo, hi, w, aligned = _linear_tables_np(int(n_frames), int(n_timesteps), rule)
        return (torch.from_numpy(lo).to(device), torch.from_numpy(hi).to(device),
                torch.from_numpy(w).to(device), aligned)
    return _cached(('lin', int(n_frames), int(n_timesteps), str(device), rule), build)


def walk_weights(n_frames, n_timesteps, device, sample_offset=0):
    """`wlin` of the frame-walking kernels, float32 [n_timesteps] on `device` (see _walk_weights_np): linear_weights with
    the next-row mark.  sample_offset > 0: a streamed piece, built on the device and not cached."""
    if not sample_offset:
        def build():
            return torch.from_numpy(_walk_full_np(int(n_frames), int(n_timesteps), RECALLED['resize'])[0]).to(device)
        return _cached(('walk', int(n_frames), int(n_timesteps), str(device), RECALLED['resize']), build)
    global _last_weights
    on_gpu = torch.device(device).type == 'cuda'
    key = ('walk', int(n_frames), int(n_timesteps), str(device), int(sample_offset), RECALLED['resize'],
           _stream().value if on_gpu else None)
    last = _last_weights
    if last is not None and last[0] == key:
        return last[1]
    w = _linear_weights_at(n_frames, n_timesteps, device, sample_offset, walk=True)
    _last_weights = (key, w)
    return w


def walkable(n_frames, n_timesteps, sample_offset=0, n=None):
    """Can the frame-walking kernels render samples sample_offset .. + n of a signal with n_frames per n_timesteps?"""
    if not sample_offset and n is None:
        return _walkable_np(int(n_frames), int(n_timesteps), RECALLED['resize'])
    return _walkable_piece_np(int(n_frames), int(n_timesteps), RECALLED['resize'], int(sample_offset),
                              int(n_timesteps if n is None else n))


def linear_weights(n_frames, n_timesteps, device, sample_offset=0):
    """Interpolation weight of every sample, float32 [n_timesteps].  sample_offset > 0: the samples are
    sample_offset, sample_offset + 1, ... of a longer signal with the same frame / sample ratio (a streamed piece,
    streaming.py): the reference forms float32(n) * scale at the ABSOLUTE n, whose fractional part rounds differently
    at different magnitudes, so a piece takes the weights the one-call render has at those positions.  Built on the
    device (exact IEEE float32 multiply / floor / subtract, the same values as the cached numpy table), not cached."""
    if not sample_offset:
        return linear_tables(n_frames, n_timesteps, device)[2]
    global _last_weights
    # the stream is part of the key: the tensor is produced on the caller's current stream and nothing orders another
    # stream's reads behind it
    on_gpu = torch.device(device).type == 'cuda'
    key = (int(n_frames), int(n_timesteps), str(device), int(sample_offset), RECALLED['resize'],
           _stream().value if on_gpu else None)
    last = _last_weights                                   # one (key, tensor) pair, read and replaced as a whole: host
    if last is not None and last[0] == key:                # threads never see one call's key with another call's tensor
        return last[1]                                     # (a streamed piece asks twice: the bank and the phase state)
    w = _linear_weights_at(n_frames, n_timesteps, device, sample_offset)
    _last_weights = (key, w)
    return w


_last_weights = None


def _linear_weights_at(n_frames, n_timesteps, device, sample_offset, walk=False):
    scale = float(F32(n_frames) / F32(n_timesteps))
    idx = torch.arange(int(sample_offset), int(sample_offset) + int(n_timesteps), device=device, dtype=torch.int64)
    n = idx.to(torch.float32)
    if RECALLED['resize'] == 'half_pixel':
        pos = (n + 0.5) * scale - 0.5
    else:
        pos = n * scale
    fl = torch.floor(pos)
    w = pos - fl
    if walk:
        # the next-row mark of _walk_weights_np, with its restrictions: a marked sample takes row t + 1 with weight 0 and lies
        # in the last block of its frame (the kernels look for marks there only).  A piece with any OTHER off-row sample --
        # hours into a signal, or under the half-pixel rule -- cannot be rendered by the frame walk: refuse it here instead
        # of rendering wrong frequencies (StreamingSynthesizer.push asks walkable() first and never gets here)
        if int(n_timesteps) % int(n_frames) != 0:
            raise ValueError(f'walk_weights: {n_timesteps} samples are not a whole number of hops of {n_frames} frames')
        u = int(n_timesteps) // int(n_frames)
        if not _walkable_piece_np(int(n_frames), int(n_timesteps), RECALLED['resize'], int(sample_offset), int(n_timesteps)):
            raise ValueError(f'walk_weights: samples {int(sample_offset)} .. {int(sample_offset) + int(n_timesteps)} of a signal '
                             f'with {u} samples per frame are not walkable under resize={RECALLED["resize"]!r} (the bilinear '
                             'resize leaves frames n // U and n // U + 1 there); render the piece through resample + '
                             'cos_oscillator_bank, or ask core.walkable() first')
        t = idx // u
        lo = fl.to(torch.int64).clamp_(min=0)
        nxt = (lo == t + 1) & (w == 0) & (idx % u >= u - WALK_BLOCK)
        w = torch.where(nxt, torch.ones_like(w), w)
    return w.contiguous()


def linear_exact_frames(upsampling):
    """Number of leading frames over which a streamed piece can reproduce the one-call render's bilinear resize: the
    reference forms float32(n) * float32(1 / U) at the absolute sample n and takes floor / fractional part.  Far enough
    into a signal that product rounds ACROSS an integer (the last sample of frame k - 1 lands on k, or the first of
    frame k below it), the one-call render then interpolates between other frames than a piece -- which indexes its
    frames locally -- can know.  Returns the first frame index at which that happens (2^24 / U if never below 2^24
    samples, where float32(n) itself stops being exact): 131 072 frames (8.7 min) at U = 96.  streaming.py refuses to
    go past it."""
    u = int(upsampling)

    def build():
        scale = F32(1.0) / F32(u)
        kmax = (1 << 24) // u
        k = np.arange(1, kmax, dtype=np.int64)
        first = np.floor((k * u).astype(F32) * scale) != k                 # first sample of frame k
        last = np.floor((k * u - 1).astype(F32) * scale) != k - 1          # last sample of frame k - 1
        bad = np.nonzero(first | last)[0]
        return int(k[bad[0]]) if bad.size else kmax
    return _cached(('linexact', u), build)


def hann_window(n, device):
    return _cached(('hann', int(n), str(device)),
                   lambda: torch.from_numpy(_hann_window_np(int(n))).to(device))


def _apply_window_rows(ir, window_size, crop_rule='ddsp370'):
    """ddsp.core.apply_window_to_impulse_response(causal=False) applied along the last axis.

    ``ir`` is float64 [..., ir_size] (zero phase); the float32 Hann window is the one TF builds.
    crop_rule: RECALLED['window_crop'] (only matters when window_size < ir_size).
    """
    ir_size = int(ir.shape[-1])
    if window_size <= 0 or window_size > ir_size:
        window_size = ir_size
    window = _hann_window_np(window_size).astype(np.float64)
    padding = ir_size - window_size
    centred = crop_rule == 'centred'
    if padding > 0:
        half_idx = window_size // 2 if centred else (window_size + 1) // 2
        window = np.concatenate([window[half_idx:], np.zeros([padding]), window[:half_idx]], axis=0)
    else:
        window = np.fft.fftshift(window)
    ir = window * ir
    if padding > 0 and centred:
        ir = np.concatenate([ir[..., ir_size - half_idx:], ir[..., :window_size - half_idx]], axis=-1)
    elif padding > 0:
        first_half_start = (ir_size - (half_idx - 1)) + 1
        second_half_end = half_idx + 1
        ir = np.concatenate([ir[..., first_half_start:], ir[..., :second_half_end]], axis=-1)
    else:
        ir = np.fft.fftshift(ir, axes=-1)
    return ir


@functools.lru_cache(maxsize=16)
def _fir_matrix_np(n_bands, window_size, crop_rule='ddsp370'):
    """M[K, Lw] with frequency_impulse_response(mag) == mag @ M (real inverse DFT x window, shifted)."""
    if n_bands < 2:
        raise ValueError('frequency_impulse_response needs at least 2 frequency bands')
    ir_size = 2 * (n_bands - 1)
    k = np.arange(n_bands, dtype=np.float64)[:, None]
    j = np.arange(ir_size, dtype=np.float64)[None, :]
    coef = np.full([n_bands, 1], 2.0)
    coef[0, 0] = 1.0
    coef[-1, 0] = 1.0
    basis = coef * np.cos(2.0 * np.pi * k * j / ir_size) / ir_size       # irfft of unit magnitudes
    return np.ascontiguousarray(_apply_window_rows(basis, int(window_size), crop_rule).astype(F32))


@functools.lru_cache(maxsize=16)
def _fir_symmetry_np(n_bands, window_size, crop_rule='ddsp370'):
    """(uniq, mirror): taps to evaluate and the tap each one is mirrored onto (-1 = none).

    The windowed zero-phase response is even, so the causal FIR satisfies ir[c + m] == ir[c - m]
    about its centre tap c; the pairing is verified numerically on the basis matrix itself."""
    m = _fir_matrix_np(n_bands, window_size, crop_rule).astype(np.float64)
    lw = m.shape[1]
    tol = 1e-6 * float(np.abs(m).max())
    for c in (lw // 2, (lw - 1) // 2):
        ok = True
        for d in range(1, lw):
            lo, hi = c - d, c + d
            if lo < 0 and hi >= lw:
                break
            if lo >= 0 and hi < lw and np.abs(m[:, lo] - m[:, hi]).max() > tol:
                ok = False
                break
        if ok:
            uniq, mirror = [], []
            for i in range(lw):
                j = 2 * c - i
                if i >= c or j >= lw:          # keep the upper half, and lower taps with no partner
                    uniq.append(i)
                    mirror.append(j if (i > c and 0 <= j < lw) else -1)
            return np.asarray(uniq, np.int32), np.asarray(mirror, np.int32)
    return np.arange(lw, dtype=np.int32), np.full([lw], -1, np.int32)


@functools.lru_cache(maxsize=16)
def _fir_eo_tables_np(n_bands, window_size):
    """Even/odd tables of the full-window FIR design, or None when the window is cropped / K unsupported.

    z[j] = E[j] + O[j], z[half - j] = E[j] - O[j] (half = K - 1); out[i] = hann[i] * z[(i + half) % Lh].
    """
    k_all = int(n_bands)
    ir_size = 2 * (k_all - 1)
    if k_all not in (32, 64, 96, 128) or (0 < window_size < ir_size):
        return None
    half = k_all - 1
    nj = half // 2 + 1
    coef = np.full([k_all], 2.0)
    coef[0] = coef[-1] = 1.0
    j = np.arange(nj, dtype=np.float64)[None, :]
    ke = np.arange(0, k_all, 2, dtype=np.float64)[:, None]
    ko = np.arange(1, k_all, 2, dtype=np.float64)[:, None]
    ce = coef[0::2, None] * np.cos(2.0 * np.pi * ke * j / ir_size) / ir_size
    co = coef[1::2, None] * np.cos(2.0 * np.pi * ko * j / ir_size) / ir_size
    win = _hann_window_np(ir_size).astype(np.float64)
    idx = np.full([nj, 4], -1, np.int32)
    we = np.zeros([nj, 4], np.float64)
    wo = np.zeros([nj, 4], np.float64)
    fill = np.zeros([nj], np.int64)
    for i in range(ir_size):
        jj = (i + half) % ir_size
        if jj > half:
            jj = ir_size - jj
        if jj <= half // 2:
            lane, sign = jj, 1.0
        else:
            lane, sign = half - jj, -1.0
        s = fill[lane]
        idx[lane, s] = i
        we[lane, s] = win[i]
        wo[lane, s] = sign * win[i]
        fill[lane] += 1
    assert fill.max() <= 4 and (idx >= 0).sum() == ir_size
    return (np.ascontiguousarray(ce.astype(F32)), np.ascontiguousarray(co.astype(F32)), idx,
            we.astype(F32), wo.astype(F32), nj, ir_size)


def fir_eo_tables(n_bands, window_size, device):
    def build():
        t = _fir_eo_tables_np(int(n_bands), int(window_size))
        if t is None:
            return None
        ce, co, idx, we, wo, nj, lw = t
        return tuple(torch.from_numpy(a).to(device) for a in (ce, co, idx, we, wo)) + (nj, lw)
    return _cached(('firEO', int(n_bands), int(window_size), str(device)), build)


def fir_matrix(n_bands, window_size, device):
    rule = RECALLED['window_crop']

    def build():
        uniq, mirror = _fir_symmetry_np(int(n_bands), int(window_size), rule)
        return (torch.from_numpy(_fir_matrix_np(int(n_bands), int(window_size), rule)).to(device),
                torch.from_numpy(uniq).to(device), torch.from_numpy(mirror).to(device))
    return _cached(('firM', int(n_bands), int(window_size), str(device), rule), build)


# ----------------------------------------------------------------------------------------------------
# upsamplers
# ----------------------------------------------------------------------------------------------------
def upsample_with_windows(inputs, n_timesteps, add_endpoint=True):
    """ddsp.core.upsample_with_windows (overlapping Hann windows)."""
    x = tf_float32(inputs)
    if x.dim() != 3:
        raise ValueError('Upsample_with_windows() only supports 3 dimensions, not {}.'.format(tuple(x.shape)))
    if not add_endpoint:
        raise NotImplementedError('add_endpoint=False is not on the DDSP-Piano path')
    n_frames = int(x.shape[1]) + 1
    n_intervals = n_frames - 1
    if n_frames >= n_timesteps:
        raise ValueError('Upsample with windows cannot be used for downsampling'
                         'More input frames ({}) than output timesteps ({})'.format(n_frames, n_timesteps))
    if n_timesteps % n_intervals != 0.0:
        raise ValueError('n_timesteps / n_intervals must be an integer')
    hop = n_timesteps // n_intervals
    b, t, c = x.shape
    y = torch.empty((b, n_timesteps, c), dtype=torch.float32, device=x.device)
    win = hann_window(2 * hop, x.device)
    _lib.check(_lib_().ddspp_resample_window(_ptr(x), _ptr(win), _ptr(y), b, t, c, hop, _stream()))
    return y


def resample(inputs, n_timesteps, method='linear', add_endpoint=True):
    """ddsp.core.resample -- call sites inharm_synth.py:117-119."""
    x = tf_float32(inputs)
    is_1d, is_2d = x.dim() == 1, x.dim() == 2
    if is_1d:
        x = x[None, :, None]
    if is_2d:
        x = x[:, :, None]
    x = x.contiguous()
    n_timesteps = int(n_timesteps)
    if method == 'linear':
        if not add_endpoint:
            raise NotImplementedError('align_corners=True resize is not on the DDSP-Piano path')
        b, t, c = x.shape
        lo, hi, w, _ = linear_tables(t, n_timesteps, x.device)
        y = torch.empty((b, n_timesteps, c), dtype=torch.float32, device=x.device)
        _lib.check(_lib_().ddspp_resample_linear(_ptr(x), _ptr(lo), _ptr(hi), _ptr(w), _ptr(y), b, t, c,
                                                 n_timesteps, _stream()))
    elif method == 'window':
        y = upsample_with_windows(x, n_timesteps, add_endpoint)
    elif method in ('nearest', 'cubic'):
        raise NotImplementedError(f"resample method '{method}' is not used by DDSP-Piano")
    else:
        raise ValueError('Method ({}) is invalid. Must be one of {}.'.format(
            method, "['nearest', 'linear', 'cubic', 'window']"))
    if is_1d:
        y = y[0, :, 0]
    if is_2d:
        y = y[:, :, 0]
    return y


# ----------------------------------------------------------------------------------------------------
# oscillator bank
# ----------------------------------------------------------------------------------------------------
def _osc_workspace(rows, n_samples, n_osc, device):
    nbytes = int(_lib_().ddspp_osc_workspace_bytes(rows, n_samples, n_osc))
    return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device), nbytes


def cos_oscillator_bank(frequency_envelopes, amplitude_envelopes, sample_rate=16000, sum_sinusoids=True,
                        use_angular_cumsum=False, spans=0):
    """ddsp_piano/modules/inharm_synth.py:49-84.  [B, N, H] envelopes -> [B, N] (or [B, N, H])."""
    fe, ae = tf_float32(frequency_envelopes), tf_float32(amplitude_envelopes)
    if fe.dim() != 3 or fe.shape != ae.shape:
        raise ValueError('frequency_envelopes {} and amplitude_envelopes {} must both be '
                         '[batch, n_samples, n_sinusoids]'.format(tuple(fe.shape), tuple(ae.shape)))
    b, n, h = fe.shape
    pad = (-n) % 8
    if pad:       # the kernel walks 8-sample blocks: zero-extend the envelopes and crop the audio
        fe = torch.nn.functional.pad(fe, (0, 0, 0, pad))
        ae = torch.nn.functional.pad(ae, (0, 0, 0, pad))
    npad = n + pad
    out = torch.empty((b, npad) if sum_sinusoids else (b, npad, h), dtype=torch.float32, device=fe.device)
    ws, nbytes = _osc_workspace(b, npad, h, fe.device)
    _lib.check(_lib_().ddspp_cos_oscillator_bank(_ptr(fe), _ptr(ae), _ptr(out), b, npad, h, float(sample_rate),
                                                 int(bool(sum_sinusoids)), int(bool(use_angular_cumsum)),
                                                 int(spans), _ptr(ws), nbytes, _stream()))
    return out[:, :n].contiguous() if pad else out


def fused_synthesis_supported(n_frames, n_samples):
    """True when harmonic_synthesis can run straight from frame controls."""
    if n_samples % n_frames != 0:
        return False
    u = n_samples // n_frames
    if u % 8 != 0 or n_frames + 1 >= n_samples:
        return False
    # (round 4: no longer "lo[n] == n // U everywhere" -- files past 131 072 frames keep the fast path, see
    # _walk_weights_np; only signals the frame walk cannot reproduce at all fall back to the three-operator route)
    return _walkable_np(int(n_frames), int(n_samples), RECALLED['resize'])


def harmonic_synthesis_fused(f0_hz, amplitudes, harmonic_distribution, harmonic_shifts, n_samples,
                             sample_rate, use_angular_cumsum, spans=0, out=None):
    """MultiInharmonic.get_signal for rows [R, T, .]: all sub-strings, envelopes never materialised."""
    r, t, s = f0_hz.shape
    h = harmonic_distribution.shape[-1]
    u = n_samples // t
    dev = f0_hz.device
    wlin = walk_weights(t, n_samples, dev)
    whann = hann_window(2 * u, dev)
    if out is None:
        out = torch.empty((r, n_samples), dtype=torch.float32, device=dev)
    ws, nbytes = _osc_workspace(r, n_samples, s * h, dev)
    _lib.check(_lib_().ddspp_harmonic_synthesis(
        _ptr(f0_hz), _ptr(amplitudes), _ptr(harmonic_distribution),
        _ptr(harmonic_shifts) if harmonic_shifts is not None else ctypes.c_void_p(0),
        _ptr(wlin), _ptr(whann), _ptr(out), r, t, s, h, u, float(sample_rate),
        int(bool(use_angular_cumsum)), int(spans), _ptr(ws), nbytes, _stream()))
    return out


def polyphonic_additive(f0_hz, amplitudes, harmonic_distribution, harmonic_shifts, n_segments, n_samples,
                        sample_rate, spans=0, voice_major=False, audible=None, split_last=False, inharm_coef=None,
                        phase_state=None, sample_offset=0, decays=None, decay_time=None):
    """Sum over the voices of each segment of MultiInharmonic.get_signal: rows [B * P, T, .] -> [B, N]
    (rows ordered [B, P], or [P, B] with voice_major=True).

    The per-voice stems are never formed; lanes go only to oscillators that are audible somewhere in a
    span (ddspp_polyphonic_additive).  Inference (angular cumsum) path only.  audible: the int32 [R, T]
    per-frame counts of InHarmonic._controls(want_counts=True) (saves a scan of the [R, T, H] controls).
    split_last=True returns (sum of voices 0 .. P-2, the last voice's stem): what the outputs dictionary of the
    reference's DAG holds for the re-used additive processor (polyphonic_dag.py:28-37).
    harmonic_shifts=None with inharm_coef [R, T] (raw): the kernels form the shifts themselves (get_inharmonic_freq).
    phase_state [R, S * H]: streaming -- the oscillators continue from the state oscillator_phase_state left;
    sample_offset: absolute position of the first sample in the streamed signal (linear_weights).
    decays [R, T, H] + decay_time [R, T]: SurrogateAdditive voices (ddspp_polyphonic_surrogate_additive; one sub-string,
    no streaming state)."""
    r, t, s = f0_hz.shape
    h = harmonic_distribution.shape[-1]
    b = int(n_segments)
    p = r // b
    u = n_samples // t
    dev = f0_hz.device
    wlin = walk_weights(t, n_samples, dev, sample_offset)
    whann = hann_window(2 * u, dev)
    lib = _lib_()
    nbytes = int(lib.ddspp_polyphonic_additive_workspace_bytes(b, p, t, s, h, u))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.empty((b, n_samples), dtype=torch.float32, device=dev)
    last = torch.empty((b, n_samples), dtype=torch.float32, device=dev) if split_last else None
    null = ctypes.c_void_p(0)
    if audible is not None and (audible.dtype != torch.int32 or audible.numel() != r * t or not audible.is_contiguous()):
        raise ValueError('audible must be a contiguous int32 tensor of R * T frame counts')
    if decays is not None:
        if s != 1 or phase_state is not None or decay_time is None:
            raise ValueError('polyphonic_additive: decays need decay_time, one sub-string and no streaming state')
        _lib.check(lib.ddspp_polyphonic_surrogate_additive(
            _ptr(f0_hz), _ptr(amplitudes), _ptr(harmonic_distribution),
            _ptr(harmonic_shifts) if harmonic_shifts is not None else null,
            _ptr(inharm_coef) if (inharm_coef is not None and harmonic_shifts is None) else null,
            ctypes.c_void_p(audible.data_ptr()) if audible is not None else null, _ptr(decays), _ptr(decay_time),
            _ptr(wlin), _ptr(whann), _ptr(out), _ptr(last), b, p, t, h, u, float(sample_rate), int(spans),
            int(bool(voice_major)), _ptr(ws), nbytes, _stream()))
        return (out, last) if split_last else out
    _lib.check(lib.ddspp_polyphonic_additive(
        _ptr(f0_hz), _ptr(amplitudes), _ptr(harmonic_distribution),
        _ptr(harmonic_shifts) if harmonic_shifts is not None else null,
        _ptr(inharm_coef) if (inharm_coef is not None and harmonic_shifts is None) else null,
        ctypes.c_void_p(audible.data_ptr()) if audible is not None else null, _ptr(wlin), _ptr(whann),
        _ptr(phase_state), _ptr(out), _ptr(last), b, p, t, s, h, u, float(sample_rate), int(spans), int(bool(voice_major)),
        _ptr(ws), nbytes, _stream()))
    return (out, last) if split_last else out


def polyphonic_stems(f0_hz, amplitudes, harmonic_distribution, harmonic_shifts, n_segments, n_samples, sample_rate,
                     spans=0, voice_major=False, audible=None, inharm_coef=None):
    """MultiInharmonic.get_signal of EVERY voice: rows [B * P, T, .] -> stems [B * P, N], rows in the order of the controls
    (ddspp_polyphonic_stems: the compacted bank of polyphonic_additive with the harmonic sum stopped at voice
    boundaries).  Inference (angular cumsum) path only; arguments as polyphonic_additive."""
    r, t, s = f0_hz.shape
    h = harmonic_distribution.shape[-1]
    b = int(n_segments)
    p = r // b
    u = n_samples // t
    dev = f0_hz.device
    wlin = walk_weights(t, n_samples, dev, 0)
    whann = hann_window(2 * u, dev)
    lib = _lib_()
    nbytes = int(lib.ddspp_polyphonic_stems_workspace_bytes(b, p, t, s, h, u))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.empty((r, n_samples), dtype=torch.float32, device=dev)
    null = ctypes.c_void_p(0)
    if audible is not None and (audible.dtype != torch.int32 or audible.numel() != r * t or not audible.is_contiguous()):
        raise ValueError('audible must be a contiguous int32 tensor of R * T frame counts')
    _lib.check(lib.ddspp_polyphonic_stems(
        _ptr(f0_hz), _ptr(amplitudes), _ptr(harmonic_distribution),
        _ptr(harmonic_shifts) if harmonic_shifts is not None else null,
        _ptr(inharm_coef) if (inharm_coef is not None and harmonic_shifts is None) else null,
        ctypes.c_void_p(audible.data_ptr()) if audible is not None else null, _ptr(wlin), _ptr(whann), _ptr(out),
        b, p, t, s, h, u, float(sample_rate), int(spans), int(bool(voice_major)), _ptr(ws), nbytes, _stream()))
    return out


def oscillator_phase_state(f0_hz, n_chunks, upsampling, sample_rate, harmonic_shifts=None, inharm_coef=None,
                           n_harmonics=None, phase_state=None, audible=None, sample_offset=0):
    """The state an oscillator bank carries across calls: phase_state [R, S * H] after the first n_chunks 1000-sample
    chunks of these controls (f0_hz [R, T, S]; harmonic_shifts [R, T, H] or raw inharm_coef [R, T]); sample_offset:
    absolute position of the controls' first sample in the streamed signal.  See ddspp_oscillator_phase_state."""
    r, t, s = f0_hz.shape
    h = int(harmonic_shifts.shape[-1]) if harmonic_shifts is not None else int(n_harmonics)
    dev = f0_hz.device
    wlin = walk_weights(t, t * int(upsampling), dev, sample_offset)
    lib = _lib_()
    nbytes = int(lib.ddspp_oscillator_phase_state_workspace_bytes(r, s, h, int(n_chunks)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.empty((r, s * h), dtype=torch.float32, device=dev)
    dummy = None
    if harmonic_shifts is None and inharm_coef is None:
        dummy = torch.zeros((r, t, h), dtype=torch.float32, device=dev)
    _lib.check(lib.ddspp_oscillator_phase_state(
        _ptr(f0_hz), _ptr(harmonic_shifts), _ptr(inharm_coef) if harmonic_shifts is None else ctypes.c_void_p(0), _ptr(dummy),
        ctypes.c_void_p(audible.data_ptr()) if audible is not None else ctypes.c_void_p(0), _ptr(wlin), _ptr(phase_state),
        _ptr(out), r, t, s, h, int(upsampling), float(sample_rate), int(n_chunks), _ptr(ws), nbytes, _stream()))
    return out


def harmonic_synthesis(frequencies, amplitudes, harmonic_shifts=None, harmonic_distribution=None,
                       n_samples=64000, sample_rate=16000, amp_resample_method='window', sum_sinusoids=True,
                       use_angular_cumsum=False):
    """ddsp_piano/modules/inharm_synth.py:87-127."""
    frequencies = tf_float32(frequencies)
    amplitudes = tf_float32(amplitudes)
    if frequencies.dim() != 3 or amplitudes.dim() != 3:
        raise ValueError('frequencies and amplitudes must be [batch, n_frames, 1]')
    b, t, _ = frequencies.shape
    if harmonic_distribution is not None:
        harmonic_distribution = tf_float32(harmonic_distribution)
        n_harmonics = int(harmonic_distribution.shape[-1])
    else:
        n_harmonics = 1
    if harmonic_shifts is not None:
        harmonic_shifts = tf_float32(harmonic_shifts)
    n_samples = int(n_samples)

    fused_ok = (sum_sinusoids and amp_resample_method == 'window' and frequencies.shape[-1] == 1
                and amplitudes.shape[-1] == 1 and fused_synthesis_supported(t, n_samples)
                and (harmonic_shifts is None or tuple(harmonic_shifts.shape) == (b, t, n_harmonics)))
    if fused_ok:
        hd = harmonic_distribution if harmonic_distribution is not None else \
            torch.ones((b, t, 1), dtype=torch.float32, device=frequencies.device)
        return harmonic_synthesis_fused(frequencies, amplitudes.reshape(b, t).contiguous(), hd,
                                        harmonic_shifts, n_samples, sample_rate, use_angular_cumsum)

    # general route: the reference's three operators, one kernel each
    harmonic_frequencies = get_harmonic_frequencies(frequencies, n_harmonics)       # :106
    if harmonic_shifts is not None:
        harmonic_frequencies = harmonic_frequencies * (1.0 + harmonic_shifts)       # :108
    if harmonic_distribution is not None:
        harmonic_amplitudes = amplitudes * harmonic_distribution                    # :112
    else:
        harmonic_amplitudes = amplitudes
    frequency_envelopes = resample(harmonic_frequencies, n_samples)                  # :117
    amplitude_envelopes = resample(harmonic_amplitudes, n_samples, method=amp_resample_method)  # :118
    return cos_oscillator_bank(frequency_envelopes, amplitude_envelopes, sample_rate=sample_rate,
                               sum_sinusoids=sum_sinusoids, use_angular_cumsum=use_angular_cumsum)


def surrogate_harmonic_synthesis(frequencies, amplitudes, decays=None, decay_time=None, harmonic_shifts=None,
                                 harmonic_distribution=None, upsampling=64, sample_rate=16000,
                                 amp_resample_method='window', use_angular_cumsum=False):
    """ddsp_piano/modules/surrogate_synth.py:11-104: the oscillator bank with a
    per-harmonic exponential decay on the amplitude envelopes."""
    frequencies, amplitudes = tf_float32(frequencies), tf_float32(amplitudes)
    b, t, _ = frequencies.shape
    n_samples = int(upsampling) * t
    if harmonic_distribution is not None:
        harmonic_distribution = tf_float32(harmonic_distribution)
        n_harmonics = int(harmonic_distribution.shape[-1])
    elif harmonic_shifts is not None:
        n_harmonics = int(tf_float32(harmonic_shifts).shape[-1])
    else:
        n_harmonics = 1
    # fused route (round 4): straight from the frame controls, the decay term inside the oscillator kernel
    # (ddspp_surrogate_harmonic_synthesis) -- no [B, N, H] envelope is formed
    if (decays is not None and decay_time is not None and amp_resample_method == 'window'
            and not _lib.options.surrogate_materialised                          # A/B switch: the three-operator route
            and frequencies.shape[-1] == 1 and amplitudes.shape[-1] == 1 and n_harmonics <= 512 and upsampling % 8 == 0
            and fused_synthesis_supported(t, n_samples)
            and (harmonic_shifts is None or tuple(harmonic_shifts.shape) == (b, t, n_harmonics))):
        dev = frequencies.device
        hd = harmonic_distribution.expand(b, t, n_harmonics).contiguous() if harmonic_distribution is not None else \
            torch.ones((b, t, n_harmonics), dtype=torch.float32, device=dev)
        sh = tf_float32(harmonic_shifts).contiguous() if harmonic_shifts is not None else None
        dec = tf_float32(decays).expand(b, t, n_harmonics).contiguous()
        dtm = tf_float32(decay_time).reshape(b, t).contiguous()
        out = torch.empty((b, n_samples), dtype=torch.float32, device=dev)
        wlin = walk_weights(t, n_samples, dev)
        whann = hann_window(2 * int(upsampling), dev)
        ws, nbytes = _osc_workspace(b, n_samples, n_harmonics, dev)
        _lib.check(_lib_().ddspp_surrogate_harmonic_synthesis(
            _ptr(frequencies.contiguous()), _ptr(amplitudes.reshape(b, t).contiguous()), _ptr(hd),
            _ptr(sh) if sh is not None else ctypes.c_void_p(0), _ptr(dec), _ptr(dtm), _ptr(wlin), _ptr(whann), _ptr(out),
            b, t, n_harmonics, int(upsampling), float(sample_rate), int(bool(use_angular_cumsum)), 0, _ptr(ws), nbytes,
            _stream()))
        return out
    harmonic_frequencies = get_harmonic_frequencies(frequencies, n_harmonics)
    if harmonic_shifts is not None:
        harmonic_frequencies = harmonic_frequencies * (1.0 + tf_float32(harmonic_shifts))
    harmonic_amplitudes = amplitudes * harmonic_distribution if harmonic_distribution is not None else amplitudes
    frequency_envelopes = resample(harmonic_frequencies, n_samples)
    amplitude_envelopes = resample(harmonic_amplitudes, n_samples, method=amp_resample_method)
    if decays is not None and decay_time is not None:
        decays = tf_float32(decays).expand(b, t, n_harmonics).contiguous()
        decay_time = tf_float32(decay_time).reshape(b, t).contiguous()
        amplitude_envelopes = amplitude_envelopes.expand(b, n_samples, n_harmonics).contiguous()
        _lib.check(_lib_().ddspp_decay_envelope(_ptr(amplitude_envelopes), _ptr(decays), _ptr(decay_time), b, t,
                                                n_harmonics, int(upsampling), _stream()))
    return cos_oscillator_bank(frequency_envelopes, amplitude_envelopes, sample_rate=sample_rate,
                               use_angular_cumsum=use_angular_cumsum)


# ----------------------------------------------------------------------------------------------------
# FilteredNoise: impulse responses and the time-varying FIR
# ----------------------------------------------------------------------------------------------------
def frequency_impulse_response(magnitudes, window_size=0, raw_scale=None):
    """ddsp.core.frequency_impulse_response: [..., K] magnitudes -> [..., Lw] causal linear-phase FIRs.

    raw_scale = (kind, bias, params) of scale_kind(): ``magnitudes`` are raw network outputs and
    scale_fn(magnitudes + bias) (FilteredNoise.get_controls) is applied inside the design kernel."""
    mags = tf_float32(magnitudes)
    k = int(mags.shape[-1])
    eo = fir_eo_tables(k, int(window_size), mags.device) if mags.data_ptr() % 16 == 0 else None
    if eo is not None:
        ce, co, idx, we, wo, nj, lw = eo
        frames = mags.numel() // k
        ir = torch.empty(tuple(mags.shape[:-1]) + (lw,), dtype=torch.float32, device=mags.device)
        if raw_scale is None:
            code, bias, prm = -1, 0.0, dict(exponent=10.0, max_value=2.0, threshold=1e-7, gain=1.0)
        else:
            code, bias, prm = raw_scale
        _lib.check(_lib_().ddspp_fir_from_magnitudes_eo(_ptr(mags), _ptr(ce), _ptr(co), _ptr(idx), _ptr(we),
                                                        _ptr(wo), _ptr(ir), frames, k, lw, nj, int(code), float(bias),
                                                        prm['exponent'], prm['max_value'], prm['threshold'],
                                                        prm['gain'], _stream()))
        return ir
    if raw_scale is not None:
        mags = scale_bias(mags, *raw_scale)
    m, uniq, mirror = fir_matrix(k, int(window_size), mags.device)
    lw = int(m.shape[1])
    frames = mags.numel() // k
    ir = torch.empty(tuple(mags.shape[:-1]) + (lw,), dtype=torch.float32, device=mags.device)
    _lib.check(_lib_().ddspp_fir_from_magnitudes(_ptr(mags), _ptr(m), _ptr(uniq), _ptr(mirror),
                                                 int(uniq.numel()), _ptr(ir), frames, k, lw, _stream()))
    return ir


def scale_bias(x, code, bias, prm):
    """scale_fn(x + bias) for the library's scale functions (ddspp_scale_bias)."""
    x = tf_float32(x)
    out = torch.empty_like(x)
    _lib.check(_lib_().ddspp_scale_bias(_ptr(x), _ptr(out), x.numel(), float(bias), int(code), prm['exponent'],
                                        prm['max_value'], prm['threshold'], prm['gain'], _stream()))
    return out


def get_fft_size(frame_size, ir_size, power_of_2=True):
    """ddsp.core.get_fft_size."""
    convolved_frame_size = ir_size + frame_size - 1
    if power_of_2:
        return int(2 ** math.ceil(math.log2(convolved_frame_size)))
    return int(convolved_frame_size)


class _PlanCache:
    """Bounded cache of library-owned rocFFT plans: least recently used plans are destroyed when the cache is full
    (whole-file synthesis with ever-changing lengths would otherwise pile plans up), plans pinned by an unfinished
    two-phase call are never evicted, and every plan carries a lock: its execution info (stream, work buffer) is set
    at enqueue time, so two host threads must not enqueue on the same plan at once."""

    def __init__(self, destroy_name, maxsize=24):
        self._destroy_name = destroy_name
        self._maxsize = maxsize
        self._entries = {}          # key -> [handle, lock, pins]
        self._order = []
        self._lock = threading.Lock()
        self._recorders = []        # lists that collect every entry handed out (CapturedGroup pins what a capture used)
        atexit.register(self.clear)

    def get(self, key, create, pin=False):
        """The entry [handle, lock, pins] of `key`.  pin=True: its pin count is raised under the cache lock, so no other
        thread can evict it between this call and the caller's use; the caller releases it with pin(entry, -1)."""
        with self._lock:
            e = self._entries.get(key)
            if e is None:
                e = self._entries[key] = [create(), threading.Lock(), 0]
                self._order.append(key)
            else:
                self._order.remove(key)
                self._order.append(key)
            if pin:
                e[2] += 1
            for rec in self._recorders:                 # pinned the moment it is recorded: an eviction between now and
                if not any(x is e for x in rec):       # the end of the capture would leave the graph a destroyed plan
                    rec.append(e)
                    e[2] += 1
            self._evict()
            return e

    def record(self):
        """Context manager: the entries handed out while it is open, each pinned once (a captured HIP graph replays the
        executions of these plans without ever calling get() again, so they must outlive the cache's LRU policy).
        Release with unpin_all(entries)."""
        cache = self

        class _Rec:
            def __enter__(self):
                self.entries = []
                with cache._lock:
                    cache._recorders.append(self.entries)
                return self.entries

            def __exit__(self, *exc):
                with cache._lock:
                    cache._recorders[:] = [r for r in cache._recorders if r is not self.entries]
        return _Rec()

    def unpin_all(self, entries):
        with self._lock:
            for e in entries:
                e[2] -= 1
            self._evict()

    def _evict(self):
        lib = None
        i = 0
        while len(self._order) > self._maxsize and i < len(self._order):
            key = self._order[i]
            e = self._entries[key]
            if e[2] > 0 or e[1].locked():
                i += 1
                continue
            lib = lib or _lib_()
            getattr(lib, self._destroy_name)(e[0])
            del self._entries[key]
            self._order.pop(i)

    def pin(self, entry, delta):
        with self._lock:
            entry[2] += delta
            if delta < 0:
                self._evict()

    def clear(self):
        try:
            lib = _lib_()
        except Exception:  # noqa: BLE001 -- interpreter / HIP runtime already going down
            return
        with self._lock:
            for e in self._entries.values():
                try:
                    getattr(lib, self._destroy_name)(e[0])
                except Exception:  # noqa: BLE001
                    pass
            self._entries.clear()
            self._order.clear()

    def __len__(self):
        return len(self._entries)


_plan_cache = _PlanCache('ddspp_fftconv_plan_destroy')


def _fftconv_plan(b, b_ir, n, l, device, key_stream=None):
    # a plan carries its stream and work buffer while it executes: one plan per (shape, stream), so that callers that
    # keep several segments in flight on different streams never share one
    key = (b, b_ir, n, l, str(device), int((_stream().value or 0) if key_stream is None else key_stream))

    def create():
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib_().ddspp_fftconv_plan_create(b, b_ir, n, l, ctypes.byref(handle)))
        return handle
    return _plan_cache.get(key, create, pin=True)        # released by the caller once its work is enqueued


def _fft_convolve_single(audio, ir, padding, delay_compensation, mask_dry=False, add_dry=False):
    b, n = audio.shape
    b_ir, l = ir.shape
    if padding == 'same':
        out_len = n
    elif padding == 'valid':
        out_len = l + n - 1
    else:
        raise ValueError('Padding must be \'valid\' or \'same\', instead of {}.'.format(padding))
    entry = _fftconv_plan(b, b_ir, n, l, audio.device)
    plan = entry[0]
    lib = _lib_()
    nbytes = int(lib.ddspp_fftconv_workspace_bytes(plan))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=audio.device)
    out = torch.empty((b, out_len), dtype=torch.float32, device=audio.device)
    try:
        with entry[1]:
            _lib.check(lib.ddspp_fftconv_execute(plan, _ptr(audio), n, _ptr(ir), _ptr(out), out_len,
                                                 _auto_delay(delay_compensation), int(mask_dry), int(add_dry), _ptr(ws),
                                                 nbytes, _stream()))
    finally:
        _plan_cache.pin(entry, -1)
    return out


def fft_convolve_prepare(batch, n_samples, ir, mask_dry=False, key_stream=None):
    """First half of the single-frame fft_convolve: transform the impulse responses [B_ir, L] on the CURRENT stream
    (which may be a side stream: the IR is known before the audio).  Returns the state fft_convolve_finish needs."""
    ir = tf_float32(ir).contiguous()
    b_ir, l = ir.shape
    entry = _fftconv_plan(int(batch), int(b_ir), int(n_samples), int(l), ir.device, key_stream=key_stream)
    plan = entry[0]
    lib = _lib_()
    nbytes = int(lib.ddspp_fftconv_workspace_bytes(plan))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=ir.device)
    try:                                             # the pin taken by _fftconv_plan is kept until fft_convolve_finish
        with entry[1]:
            _lib.check(lib.ddspp_fftconv_transform_ir(plan, _ptr(ir), int(mask_dry), _ptr(ws), nbytes, _stream()))
    except Exception:
        _plan_cache.pin(entry, -1)
        raise
    return {'plan': plan, 'entry': entry, 'ws': ws, 'nbytes': nbytes, 'n': int(n_samples), 'l': int(l), 'batch': int(batch),
            'ir': ir}


def fft_convolve_finish(state, audio, padding='same', delay_compensation=-1, add_dry=False):
    """Second half: audio [B, N] against the prepared impulse-response spectra, on the current stream (the caller has
    made it wait for the stream fft_convolve_prepare ran on)."""
    audio = tf_float32(audio)
    b, n = audio.shape
    if (b, n) != (state['batch'], state['n']):
        raise ValueError(f'audio {tuple(audio.shape)} does not match the prepared {(state["batch"], state["n"])}')
    out_len = n if padding == 'same' else state['l'] + n - 1
    out = torch.empty((b, out_len), dtype=torch.float32, device=audio.device)
    entry = state.pop('entry', None)
    try:
        with (entry[1] if entry is not None else threading.Lock()):
            _lib.check(_lib_().ddspp_fftconv_execute_prepared(state['plan'], _ptr(audio), n, _ptr(out), out_len,
                                                              _auto_delay(delay_compensation), int(add_dry),
                                                              _ptr(state['ws']), state['nbytes'], _stream()))
    finally:
        if entry is not None:
            _plan_cache.pin(entry, -1)
    return out


def fft_convolve(audio, impulse_response, padding='same', delay_compensation=-1):
    """ddsp.core.fft_convolve -- audio [B, N]; impulse_response [B, L] or [B, F, L].

    F == 1 goes through rocFFT exactly as the reference does (fft_size = 2**ceil(log2(N + L - 1)));
    F > 1 (one FIR per audio frame, the FilteredNoise case) is the direct time-varying FIR kernel,
    equal to the reference's framed FFT convolution + overlap-add up to float32 round-off.
    """
    audio, ir = tf_float32(audio), tf_float32(impulse_response)
    if audio.dim() != 2:
        raise ValueError('audio must be [batch, n_samples]')
    if ir.dim() == 2:
        ir = ir[:, None, :]
    if ir.dim() != 3:
        raise ValueError('impulse_response must be [batch, ir_size] or [batch, n_frames, ir_size]')
    batch_size_ir, n_ir_frames, ir_size = ir.shape
    batch_size, audio_size = audio.shape
    if batch_size_ir != batch_size and not (batch_size_ir == 1 and batch_size > 1):
        raise ValueError('Batch size of audio ({}) and impulse response ({}) must be the same.'.format(
            batch_size, batch_size_ir))
    frame_size = int(math.ceil(audio_size / n_ir_frames))
    n_audio_frames = int(math.ceil(audio_size / frame_size))
    if n_audio_frames != n_ir_frames:
        raise ValueError('Number of Audio frames ({}) and impulse response frames ({}) do not match. '
                         'For small hop size = ceil(audio_size / n_ir_frames), number of impulse '
                         'response frames must be a multiple of the audio size.'.format(
                             n_audio_frames, n_ir_frames))
    if padding not in ('same', 'valid'):
        raise ValueError('Padding must be \'valid\' or \'same\', instead of {}.'.format(padding))
    if n_ir_frames == 1:
        return _fft_convolve_single(audio, ir[:, 0, :].contiguous(), padding, delay_compensation)
    if padding != 'same':
        raise NotImplementedError("framed fft_convolve supports padding='same' only")
    if batch_size_ir == 1 and batch_size > 1:
        ir = ir.expand(batch_size, -1, -1)
    ir = ir.contiguous()
    padded = n_ir_frames * frame_size
    x = audio if padded == audio_size else torch.nn.functional.pad(audio, (0, padded - audio_size))
    x = x.contiguous()
    out = torch.empty((batch_size, padded), dtype=torch.float32, device=audio.device)
    _lib.check(_lib_().ddspp_time_varying_fir(_ptr(x), _ptr(ir), _ptr(out), batch_size, padded, n_ir_frames,
                                              ir_size, _auto_delay(delay_compensation), _stream()))
    return out if padded == audio_size else out[:, :audio_size].contiguous()


def frequency_filter_voice_sums(audio, magnitudes, window_size, raw_scale, n_voices, voices_per_row, voice_major,
                                split_last=False):
    """frequency_filter over the rows of a polyphonic group with `voices_per_row` consecutive voices of a segment
    summed into one output row ([R / voices_per_row, N], segment major); None when the fused kernel does not apply.
    split_last=True: returns (sums, last) -- every segment's last voice [B, N] stays out of its row's sum."""
    return _frequency_filter_fused(audio, magnitudes, window_size, 'same', raw_scale,
                                   voices=(int(n_voices), int(voices_per_row), bool(voice_major), bool(split_last)))


def frequency_filter(audio, magnitudes, window_size=0, padding='same', raw_scale=None):
    """ddsp.core.frequency_filter -- call site filtered_noise_synth.py:41-42."""
    fused = _frequency_filter_fused(audio, magnitudes, window_size, padding, raw_scale)
    if fused is not None:
        return fused
    if isinstance(audio, DrawnNoise):
        audio = audio.materialise()
    impulse_response = frequency_impulse_response(magnitudes, window_size=window_size, raw_scale=raw_scale)
    return fft_convolve(audio, impulse_response, padding=padding)


class DrawnNoise:
    """U(-1, 1) noise [rows, n] that has NOT been drawn yet: the (seed, offset) of the library's Philox4x32-10 stream that
    uniform_noise((rows, n), seed, offset) would draw it from.  frequency_filter / frequency_filter_voice_sums hand it to the
    windowed kernel, which draws the numbers while staging them (ddspp_frequency_filter_eo_voices_drawn, round 6: the
    [rows, n] tensor is never written or read back); any other route calls materialise() -- the same numbers, bit for bit."""

    def __init__(self, rows, n, seed, offset, device):
        self.shape = (int(rows), int(n))
        self.seed, self.offset, self.device = int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), device

    def materialise(self):
        return uniform_noise(self.shape, seed=self.seed, offset=self.offset, device=self.device)


def _frequency_filter_fused(audio, magnitudes, window_size, padding, raw_scale, voices=None):
    """FIR design + time-varying FIR in one kernel (ddspp_frequency_filter_eo) when the shape fits, else None.
    Bit-identical to the two-kernel form; the [B, T, Lw] impulse responses are never materialised."""
    drawn = audio if isinstance(audio, DrawnNoise) else None
    if padding != 'same' or not ((drawn is not None or torch.is_tensor(audio)) and torch.is_tensor(magnitudes)):
        return None
    mags = tf_float32(magnitudes)
    if drawn is not None:
        # the shape decides first: only the windowed kernel draws its own noise
        if (mags.dim() != 3 or drawn.shape[0] != mags.shape[0] or drawn.shape[1] % 4 or mags.shape[1] < 2 or
                drawn.shape[1] % mags.shape[1]):
            audio, drawn = drawn.materialise(), None
        else:
            eo = fir_eo_tables(int(mags.shape[2]), int(window_size), mags.device)
            if eo is None or not _lib_().ddspp_frequency_filter_eo_drawn_supported(
                    drawn.shape[1], int(mags.shape[1]), int(mags.shape[2]), eo[6], _auto_delay(-1)):
                audio, drawn = drawn.materialise(), None
    if drawn is not None:
        mags = mags.contiguous()
        if mags.data_ptr() % 16:
            return None
        b, n = drawn.shape
        t, k = int(mags.shape[1]), int(mags.shape[2])
        ce, co, idx, we, wo, nj, lw = fir_eo_tables(k, int(window_size), mags.device)
        code, bias, prm = (-1, 0.0, dict(exponent=10.0, max_value=2.0, threshold=1e-7, gain=1.0)) if raw_scale is None else raw_scale
        n_voices, vq, vmajor, split_last = voices if voices is not None else (1, 1, False, False)
        if n_voices % vq or b % n_voices or (split_last and vq < 2):
            return None
        out = torch.empty((b // vq, n), dtype=torch.float32, device=mags.device)
        last = torch.empty((b // n_voices, n), dtype=torch.float32, device=mags.device) if split_last else None
        _lib.check(_lib_().ddspp_frequency_filter_eo_voices_drawn(
            drawn.seed, drawn.offset, _ptr(mags), _ptr(ce), _ptr(co), _ptr(idx), _ptr(we), _ptr(wo), _ptr(out), _ptr(last), b, n,
            t, k, lw, nj, _auto_delay(-1), int(code), float(bias), prm['exponent'], prm['max_value'], prm['threshold'],
            prm['gain'], n_voices, vq, int(vmajor), _stream()))
        return (out, last) if split_last else out
    x = tf_float32(audio)
    if x.dim() != 2 or mags.dim() != 3 or x.shape[0] != mags.shape[0]:
        return None
    b, n = x.shape
    t, k = int(mags.shape[1]), int(mags.shape[2])
    if t < 2 or n % t != 0:
        return None
    eo = fir_eo_tables(k, int(window_size), mags.device)
    if eo is None:
        return None
    ce, co, idx, we, wo, nj, lw = eo
    lib = _lib_()
    dcode = _auto_delay(-1)
    if not lib.ddspp_frequency_filter_eo_supported(n, t, k, lw, dcode):
        return None
    x, mags = x.contiguous(), mags.contiguous()
    if x.data_ptr() % 16 or mags.data_ptr() % 16:
        return None
    if raw_scale is None:
        code, bias, prm = -1, 0.0, dict(exponent=10.0, max_value=2.0, threshold=1e-7, gain=1.0)
    else:
        code, bias, prm = raw_scale
    if voices is None:
        out = torch.empty((b, n), dtype=torch.float32, device=x.device)
        _lib.check(lib.ddspp_frequency_filter_eo(_ptr(x), _ptr(mags), _ptr(ce), _ptr(co), _ptr(idx), _ptr(we), _ptr(wo),
                                                 _ptr(out), b, n, t, k, lw, nj, dcode, int(code), float(bias),
                                                 prm['exponent'], prm['max_value'], prm['threshold'], prm['gain'],
                                                 _stream()))
        return out
    n_voices, vq, vmajor, split_last = voices
    if n_voices % vq or b % n_voices or (split_last and vq < 2):
        return None
    out = torch.empty((b // vq, n), dtype=torch.float32, device=x.device)
    last = torch.empty((b // n_voices, n), dtype=torch.float32, device=x.device) if split_last else None
    _lib.check(lib.ddspp_frequency_filter_eo_voices(
        _ptr(x), _ptr(mags), _ptr(ce), _ptr(co), _ptr(idx), _ptr(we), _ptr(wo), _ptr(out), _ptr(last), b, n, t, k, lw, nj,
        dcode, int(code), float(bias), prm['exponent'], prm['max_value'], prm['threshold'], prm['gain'], n_voices, vq,
        int(vmajor), _stream()))
    return (out, last) if split_last else out


def uniform_noise(shape, seed=0, offset=0, device=None, out=None):
    """U(-1, 1) noise from the library's Philox4x32-10 (stand-in for the unseeded tf.random.uniform).  out: a contiguous
    float32 tensor of that many elements (a multiple of 4) to draw into."""
    n = int(np.prod(shape))
    if out is not None:
        if out.numel() != n or n % 4 or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError('uniform_noise(out=): a contiguous float32 tensor of prod(shape) elements, a multiple of 4')
        _lib.check(_lib_().ddspp_uniform_noise(_ptr(out), n, int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), _stream()))
        return out.view(shape)
    device = device or default_device()
    npad = (n + 3) // 4 * 4
    out = torch.empty(npad, dtype=torch.float32, device=device)
    _lib.check(_lib_().ddspp_uniform_noise(_ptr(out), npad, int(seed) & (2 ** 64 - 1),
                                           int(offset) & (2 ** 64 - 1), _stream()))
    return out[:n].reshape(shape)


def add_signals(signals):
    """MultiAdd.get_signal / processors.Add: ((s0 + s1) + s2) ... in one pass."""
    sigs = [tf_float32(s) for s in signals]
    if len(sigs) == 1:
        return sigs[0]
    shape = torch.broadcast_shapes(*[s.shape for s in sigs])
    sigs = [s.expand(shape).contiguous() for s in sigs]
    n = sigs[0].numel()
    if n % 4 != 0:
        out = sigs[0]
        for s in sigs[1:]:
            out = out + s
        return out
    ptrs = torch.tensor([s.data_ptr() for s in sigs], dtype=torch.int64, device=sigs[0].device)
    out = torch.empty(shape, dtype=torch.float32, device=sigs[0].device)
    _lib.check(_lib_().ddspp_add_signals(_ptr(ptrs), len(sigs), _ptr(out), n, _stream()))
    return out
