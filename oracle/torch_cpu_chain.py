"""Op-by-op torch-CPU restatement of the synthesis chain (TEST / BASELINE INFRASTRUCTURE ONLY).

SURVEY.md 8(d) asks for a CPU baseline "with all cores" next to the numpy oracle: this module evaluates the
same operator sequence the TF/ddsp reference runs eagerly on a CPU -- one vectorised library call per reference
op, full [B, N, H] envelopes materialised, framed FFT convolution for the noise, one big FFT for the reverb --
with torch's intra-op thread pool standing in for TensorFlow's.  Function by function it follows
oracle/ddsp_oracle.py (which cites the reference file:line of every step); it is NOT float32-order faithful
(torch.cumsum is free to block its scan), so it is a timing baseline and a loose numerical cross-check, never a
parity oracle.  Only bench.py's cpu_baseline leg and tests/ import it.
"""
from __future__ import annotations

import math

import torch

TWO_PI = 6.2831855


def exp_sigmoid(x, exponent=10.0, max_value=2.0, threshold=1e-7):
    return max_value * torch.sigmoid(x) ** math.log(exponent) + threshold


def get_controls(amplitudes, harmonic_distribution, inharm_coef, f0_hz, sample_rate, min_frequency=20.0):
    """MultiInharmonic.get_controls with the default flags (inharm_synth.py:167-219, :254-270)."""
    n_sub = f0_hz.shape[-1]
    f0 = f0_hz[..., 0:1]
    inharm_coef = torch.clamp(inharm_coef, min=0.0)
    amp = exp_sigmoid(amplitudes)
    hd = exp_sigmoid(harmonic_distribution)
    h = hd.shape[-1]
    k = torch.linspace(1.0, float(h), h)[None, None, :]
    factor = torch.sqrt(k * k * inharm_coef + 1.0)
    freq = f0 * k * factor
    shifts = factor - 1.0
    hd = torch.where(freq >= sample_rate / 2.0, torch.zeros_like(hd), hd)
    amp = amp * (f0 > min_frequency).to(amp.dtype)
    den = hd.sum(-1, keepdim=True)
    hd = hd / torch.where(den == 0.0, torch.full_like(den, 1e-7), den)
    return amp / n_sub, hd, shifts


def resample_linear(x, n):
    t = x.shape[1]
    scale = torch.tensor(float(t), dtype=torch.float32) / torch.tensor(float(n), dtype=torch.float32)
    pos = torch.arange(n, dtype=torch.float32) * scale
    lo = pos.floor().long()
    hi = torch.clamp(pos.ceil().long(), max=t - 1)
    w = (pos - pos.floor())[None, :, None]
    top, bot = x[:, lo, :], x[:, hi, :]
    return top + (bot - top) * w


def resample_window(x, n):
    t = x.shape[1]
    u = n // t
    win = torch.hann_window(2 * u, periodic=True, dtype=torch.float32)
    nxt = torch.cat([x[:, 1:], x[:, -1:]], dim=1)
    a0 = x.repeat_interleave(u, dim=1)
    a1 = nxt.repeat_interleave(u, dim=1)
    r = torch.arange(n) % u
    return a0 * win[u + r][None, :, None] + a1 * win[r][None, :, None]


def angular_cumsum(omega, chunk=1000):
    b, n, h = omega.shape
    pad = (-n) % chunk
    if pad:
        omega = torch.nn.functional.pad(omega, (0, 0, 0, pad))
    c = omega.reshape(b, -1, chunk, h)
    phase = torch.cumsum(c, dim=2)
    off = torch.remainder(phase[:, :, -1:, :], TWO_PI)
    off = torch.cat([torch.zeros_like(off[:, :1]), off[:, :-1]], dim=1)
    off = torch.remainder(torch.cumsum(off, dim=1), TWO_PI)
    phase = torch.remainder(phase + off, TWO_PI)
    return phase.reshape(b, -1, h)[:, :n]


def additive(amp, hd, shifts, f0_hz, n, sample_rate):
    audio = None
    h = hd.shape[-1]
    k = torch.linspace(1.0, float(h), h)[None, None, :]
    for s in range(f0_hz.shape[-1]):
        hf = f0_hz[..., s:s + 1] * k * (1.0 + shifts)
        ha = amp * hd
        fe = resample_linear(hf, n)
        ae = resample_window(ha, n)
        ae = torch.where(fe >= sample_rate / 2.0, torch.zeros_like(ae), ae)
        omega = fe * TWO_PI / float(sample_rate)
        sig = (ae * torch.cos(angular_cumsum(omega))).sum(-1)
        audio = sig if audio is None else audio + sig
    return audio


def frequency_filter(noise, magnitudes, window_size=257):
    """ddsp.core.frequency_filter: per-frame FIR design + framed FFT convolution + overlap-add."""
    b, n = noise.shape
    t, kk = magnitudes.shape[1], magnitudes.shape[2]
    ir = torch.fft.irfft(magnitudes.to(torch.complex64), dim=-1)
    ir_size = ir.shape[-1]
    if window_size <= 0 or window_size > ir_size:
        window_size = ir_size
    win = torch.hann_window(window_size, periodic=True, dtype=torch.float32)
    padding = ir_size - window_size
    if padding > 0:
        half = (window_size + 1) // 2
        win = torch.cat([win[half:], torch.zeros(padding), win[:half]])
        ir = win * ir
        ir = torch.cat([ir[..., (ir_size - (half - 1)) + 1:], ir[..., :half + 1]], dim=-1)
    else:
        ir = torch.fft.fftshift(torch.fft.fftshift(win) * ir, dim=-1)
    lw = ir.shape[-1]
    frame = n // t
    nfft = 2 ** math.ceil(math.log2(lw + frame - 1))
    frames = noise.reshape(b, t, frame)
    y = torch.fft.irfft(torch.fft.rfft(frames, nfft) * torch.fft.rfft(ir, nfft), nfft)
    out_len = (t - 1) * frame + nfft
    ola = torch.nn.functional.fold(y.transpose(1, 2), (1, out_len), (1, nfft), stride=(1, frame)).reshape(b, out_len)
    start = (lw - 1) // 2 - 1
    return ola[:, start:start + n]


def reverb(audio, ir):
    b, n = audio.shape
    ir = torch.cat([torch.zeros_like(ir[:, :1]), ir[:, 1:]], dim=1)
    nfft = 2 ** math.ceil(math.log2(n + ir.shape[1] - 1))
    wet = torch.fft.irfft(torch.fft.rfft(audio, nfft) * torch.fft.rfft(ir, nfft), nfft)[:, :n]
    return wet + audio


def synthesize(voices, reverb_ir, noises, sample_rate, frame_rate=250):
    """voices: list of dicts (amplitudes, harmonic_distribution, inharm_coef, f0_hz, magnitudes: torch [B, T, .]);
    noises: list of [B, N]; returns the reverberated mix [B, N] (polyphonic_dag.py:24-40)."""
    u = sample_rate // frame_rate
    mix = None
    for v, z in zip(voices, noises):
        t = v['f0_hz'].shape[1]
        n = t * u
        amp, hd, shifts = get_controls(v['amplitudes'], v['harmonic_distribution'], v['inharm_coef'], v['f0_hz'],
                                       sample_rate)
        a = additive(amp, hd, shifts, v['f0_hz'], n, sample_rate)
        zf = frequency_filter(z, exp_sigmoid(v['magnitudes'] - 5.0))
        mix = (zf + a) if mix is None else ((mix + zf) + a)
    return reverb(mix, reverb_ir)
