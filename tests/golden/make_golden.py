#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the BUILD container, where /root/reference exists).

    python tests/golden/make_golden.py

Writes (all small, float32 .npz):
  dafx22_reverb_ir.npz   rows 0 and 9 of the learned reverb bank [10, 24000] of the reference's dafx22
                         checkpoint (ddsp_piano/model_weights/dafx22/ckpt-0.data-00000-of-00001, raw
                         little-endian float32 at byte offset 308892, located by decoding ckpt-0.index;
                         SURVEY.md fact 6).  This is DATA shipped with the reference, not source.
  c1_mono.npz            BASELINE config 1: 1 s monophonic note, poly=1, 64 harmonics, 24 kHz, dry.
  c2_small.npz           down-sized config 2: B=1, P=2, T=50, H=96, K=64, S=2, 16 kHz, full chain with
                         the real dafx22 IR (first 6000 taps of row 0).
  recalled_details.npz   one tiny single-operator case per recalled ddsp detail (oracle.RECALLED): flat-spectrum
                         frequency_filter through the full and the cropped window, a resampled ramp, the angular
                         cumsum of a constant, exp_sigmoid and the FilteredNoise bias.
Every file carries a `backend` field.  "restatement": the expected outputs come from oracle/ddsp_oracle.py,
because TensorFlow / ddsp cannot be imported in the build container.  On a host with
TensorFlow + ddsp 3.7.0 + a checkout of the reference,

    DDSP_GOLDEN_BACKEND=tf [DDSP_PIANO_REFERENCE=/path/to/ddsp-piano] python tests/golden/make_golden.py

runs the SAME inputs through the real library (tests/golden/tf_backend.py), rewrites the files with
backend = "tf", and prints, per recalled detail, which oracle setting the real outputs match.  From then on
tests/test_golden_backend.py (CPU) holds the oracle, and the -m gpu golden tests hold the HIP path, to reference
outputs: parity is pinned.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import ddsp_oracle as O  # noqa: E402
from util import synth_controls  # noqa: E402

REF_CKPT = '/root/reference/ddsp_piano/model_weights/dafx22/ckpt-0.data-00000-of-00001'
IR_OFFSET, IR_ROWS, IR_LEN = 308892, 10, 24000


def backend():
    """(module-like object with the oracle's constructors, backend name, version string)."""
    if os.environ.get('DDSP_GOLDEN_BACKEND') == 'tf':
        sys.path.insert(0, HERE)
        from tf_backend import TFBackend
        b = TFBackend()
        return b, 'tf', ', '.join(f'{k} {v}' for k, v in b.versions.items())
    return O, 'restatement', f'numpy {np.__version__}'


def recalled_cases(B_):
    """Inputs and the backend's outputs for the single-operator cases that decide each recalled detail."""
    rng = np.random.default_rng(77)
    noise = rng.uniform(-1, 1, [1, 960]).astype(np.float32)
    ramp = np.arange(12, dtype=np.float32)[None, :, None]
    omega = np.full([1, 2500, 1], 0.01, np.float32)
    x = np.linspace(-6, 6, 25).astype(np.float32)
    raw_mag = rng.normal(0, 1, [1, 4, 8]).astype(np.float32)
    out = dict(noise=noise, ramp=ramp, omega=omega, x=x, raw_mag=raw_mag)
    out['flat_full'] = B_.frequency_filter(noise, np.ones([1, 10, 96], np.float32), 257)      # auto_delay
    out['flat_crop'] = B_.frequency_filter(noise, np.ones([1, 10, 200], np.float32), 257)     # window_crop + auto_delay
    out['ramp_linear'] = B_.resample(ramp, 12 * 96)                                            # resize
    out['ramp_window'] = B_.resample(ramp, 12 * 96, method='window')
    out['phase'] = B_.angular_cumsum(omega)                                                    # angular_cumsum
    # ... and the case that can tell the variants of angular_cumsum apart: 301 chunks, three
    # oscillators -- omega near pi (a partial just below Nyquist), near 0, and in between -- piecewise constant over 250
    # samples.  After 300 chunks the running offset sum is ~2 pi x 150: `phase + offsets` is rounded at ulp(940) = 6e-5 rad
    # under 'plain' offsets and at ulp(2 pi) under 'wrapped' ones, and float32 scans are deterministic, so the stored
    # phases are compared BITWISE.  Kept: every 41st sample and the whole last chunk.
    rng2 = np.random.default_rng(78)         # (a generator of its own: the draws of the older cases stay what they were)
    blocks = np.stack([rng2.uniform(2.9, 3.14, 1204), rng2.uniform(1e-5, 1e-3, 1204), rng2.uniform(0.05, 0.5, 1204)], -1)
    # (the values are omegas an oscillator bank can be driven to: omega = fl(fl(fe * fl(2 pi)) / 24000) of float32
    # frequencies fe, inharm_synth.py:69-70, so the GPU test feeds `fe_long_blocks` to cos_oscillator_bank and lands on
    # exactly these phases)
    fe_blocks = (blocks * (24000.0 / (2.0 * np.pi))).astype(np.float32)
    blocks = ((fe_blocks * O.TWO_PI_F32).astype(np.float32) / np.float32(24000.0)).astype(np.float32)
    omega_long = np.repeat(blocks, 250, axis=0)[None]                                          # [1, 301000, 3]
    out['fe_long_blocks'] = fe_blocks
    out['omega_long'] = omega_long
    ph = np.asarray(B_.angular_cumsum(omega_long), np.float32)
    out['phase_long_strided'] = ph[:, ::41]
    out['phase_long_tail'] = ph[:, -1000:]
    # bitwise cases for the two upsamplers at a hop that is NOT a power of two (w = frac(float32(n) * float32(T / N)) is
    # quantised by the product) and past the end-point handling of both
    rs_in = rng2.normal(0, 1, [2, 37, 3]).astype(np.float32)
    out['rs_in'] = rs_in
    out['rs_linear_96'] = B_.resample(rs_in, 37 * 96)
    out['rs_linear_nonint'] = B_.resample(rs_in, 1000)                                         # N % T != 0
    out['rs_window_96'] = B_.resample(rs_in, 37 * 96, method='window')
    out['exp_sigmoid'] = B_.exp_sigmoid(x)                                                     # exp_sigmoid constants
    out['noise_controls'] = B_.FilteredNoise(frame_rate=250, sample_rate=24000).get_controls(raw_mag)['magnitudes']
    # ---- not switches, but recalled all the same: one run on a TF host settles these too
    # framed fft_convolve whose audio length is NOT a multiple of the frame count (frame = ceil(1000 / 7) = 143, padded)
    fc_audio = rng.uniform(-1, 1, [1, 1000]).astype(np.float32)
    fc_ir = (rng.normal(0, 1, [1, 7, 33]) * np.hanning(33)[None, None, :]).astype(np.float32)
    out.update(fc_audio=fc_audio, fc_ir=fc_ir)
    out['fc_same'] = B_.fft_convolve(fc_audio, fc_ir, 'same', -1)
    out['fc_valid_delay0'] = B_.fft_convolve(fc_audio, fc_ir, 'valid', 0)
    # ddsp.effects.Reverb: the dry tap ir[:, 0] is masked, add_dry adds the input back (and add_dry=False does not)
    rv_audio = rng.normal(0, 0.1, [2, 700]).astype(np.float32)
    rv_ir = (rng.normal(0, 1, [2, 300]) * np.exp(-np.arange(300) / 60.0)[None, :]).astype(np.float32)
    rv_ir[:, 0] = 5.0
    out.update(rv_audio=rv_audio, rv_ir=rv_ir)
    out['rv_wet_dry'] = B_.Reverb(add_dry=True).get_signal(audio=rv_audio, ir=rv_ir)
    out['rv_wet'] = B_.Reverb(add_dry=False).get_signal(audio=rv_audio, ir=rv_ir)
    # upsample_with_windows: both ends of a short sequence (the added end point, the cropped half windows)
    up_in = rng.normal(0, 1, [1, 5, 3]).astype(np.float32)
    out['up_in'] = up_in
    out['up_window'] = B_.resample(up_in, 5 * 32, method='window')
    # the reference's own exp_tanh (inharm_synth.py:13-17) and MultiAdd's summation order (inharm_synth.py:296-309)
    out['exp_tanh'] = B_.exp_tanh(x)
    ma = np.stack([np.full([1, 64], 3.0e7, np.float32), rng.normal(0, 1, [1, 64]).astype(np.float32),
                   np.full([1, 64], -3.0e7, np.float32), rng.normal(0, 1, [1, 64]).astype(np.float32)])
    out['ma_in'] = ma
    out['ma_sum'] = B_.multi_add(list(ma))
    return {k: np.asarray(v, np.float32) for k, v in out.items()}


def report_recalled(cases):
    """Which oracle setting reproduces the backend's outputs (meaningful when the backend is the real library)."""
    def err(a, b):
        return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))
    ones96, ones200 = np.ones([1, 10, 96], np.float32), np.ones([1, 10, 200], np.float32)
    rows = []
    for rule in O.RECALLED_CHOICES['auto_delay']:
        with O.recalled(auto_delay=rule):
            rows.append(('auto_delay', rule, err(O.frequency_filter(cases['noise'], ones96, 257), cases['flat_full'])))
    for crop in O.RECALLED_CHOICES['window_crop']:
        for rule in O.RECALLED_CHOICES['auto_delay']:
            with O.recalled(window_crop=crop, auto_delay=rule):
                rows.append((f'window_crop (auto_delay={rule})', crop,
                             err(O.frequency_filter(cases['noise'], ones200, 257), cases['flat_crop'])))
    for rule in O.RECALLED_CHOICES['resize']:
        with O.recalled(resize=rule):
            rows.append(('resize', rule, err(O.resample(cases['ramp'], 12 * 96), cases['ramp_linear'])))
    for rule in O.RECALLED_CHOICES['angular_cumsum']:
        with O.recalled(angular_cumsum=rule):
            rows.append(('angular_cumsum', rule, err(O.angular_cumsum(cases['omega']), cases['phase'])))
    # the long case decides both scan details at once, bitwise (float32 scans are deterministic; TF's CPU cumsum is
    # sequential per lane -- if the real library shows a few-ulp cloud around ONE of the four variants instead of an exact
    # match, its scan order differs from the sequential one and that variant is still the one)
    for scan in O.RECALLED_CHOICES['angular_cumsum']:
        for offs in O.RECALLED_CHOICES['angular_offsets']:
            with O.recalled(angular_cumsum=scan, angular_offsets=offs):
                ph = O.angular_cumsum(cases['omega_long'])
            got = np.concatenate([ph[:, ::41].ravel(), ph[:, -1000:].ravel()])
            want = np.concatenate([cases['phase_long_strided'].ravel(), cases['phase_long_tail'].ravel()])
            d = np.abs(np.angle(np.exp(1j * (got.astype(np.float64) - want.astype(np.float64)))))
            rows.append((f'angular_offsets (scan={scan})', offs, float(np.sqrt(np.mean(d ** 2))),
                         f'{int((got != want).sum())} of {got.size} phases differ bitwise, max {d.max():.2e} rad'))
    # ... and whether angular_cumsum wraps what it returns (round 6): the stored phases ARE the function's output, so a TF
    # golden in [0, 2 pi) settles 'final', one that grows to thousands of radians settles 'none' -- no tolerance involved
    for wrap in O.RECALLED_CHOICES['angular_wrap']:
        with O.recalled(angular_wrap=wrap):
            ph = O.angular_cumsum(cases['omega_long'])
        got = np.concatenate([ph[:, ::41].ravel(), ph[:, -1000:].ravel()])
        want = np.concatenate([cases['phase_long_strided'].ravel(), cases['phase_long_tail'].ravel()])
        rows.append(('angular_wrap', wrap, err(got, want), f'golden phases span [{want.min():.3g}, {want.max():.6g}] rad'))
    for key, fn in (('rs_linear_96', lambda: O.resample(cases['rs_in'], 37 * 96)),
                    ('rs_linear_nonint', lambda: O.resample(cases['rs_in'], 1000)),
                    ('rs_window_96', lambda: O.resample(cases['rs_in'], 37 * 96, method='window'))):
        got = fn()
        rows.append((f'bitwise: {key}', '-', err(got, cases[key]),
                     f'{int((got != cases[key]).sum())} of {got.size} values differ bitwise'))
    rows.append(('exp_sigmoid', str(O.RECALLED['exp_sigmoid']), err(O.exp_sigmoid(cases['x']), cases['exp_sigmoid'])))
    rows.append(('initial_bias', str(O.RECALLED['initial_bias']),
                 err(O.FilteredNoise().get_controls(cases['raw_mag'])['magnitudes'], cases['noise_controls'])))
    rows.append(('framed fft_convolve, N % T != 0', "'same', auto delay", err(O.fft_convolve(cases['fc_audio'], cases['fc_ir'], 'same', -1), cases['fc_same'])))
    rows.append(('framed fft_convolve, N % T != 0', "'valid', delay 0", err(O.fft_convolve(cases['fc_audio'], cases['fc_ir'], 'valid', 0), cases['fc_valid_delay0'])))
    rows.append(('Reverb dry mask, add_dry=True', '-', err(O.Reverb(add_dry=True).get_signal(cases['rv_audio'], cases['rv_ir']), cases['rv_wet_dry'])))
    rows.append(('Reverb dry mask, add_dry=False', '-', err(O.Reverb(add_dry=False).get_signal(cases['rv_audio'], cases['rv_ir']), cases['rv_wet'])))
    rows.append(('upsample_with_windows end points', '-', err(O.resample(cases['up_in'], 5 * 32, method='window'), cases['up_window'])))
    rows.append(('exp_tanh', '-', err(O.exp_tanh(cases['x']), cases['exp_tanh'])))
    rows.append(('MultiAdd order', '((s0+s1)+s2)+s3', err(O.multi_add(list(cases['ma_in'])), cases['ma_sum'])))
    print('recalled detail                      oracle setting        rms error vs backend output')
    for name, rule, e, *note in rows:
        print(f'{name:36s} {rule:20s} {e:.3e}' + ('   <-- matches' if e < 1e-5 else '') + (f'   [{note[0]}]' if note else ''))


def dafx22_ir():
    with open(REF_CKPT, 'rb') as f:
        f.seek(IR_OFFSET)
        bank = np.frombuffer(f.read(IR_ROWS * IR_LEN * 4), dtype='<f4').reshape(IR_ROWS, IR_LEN)
    assert abs(bank[0, 1] - 3.18) < 0.05, bank[0, :3]      # first taps ~ [-7.8e-5, 3.18, -1.3e-2]
    return np.ascontiguousarray(bank[[0, 9]])


def main():
    global HERE
    committed = HERE
    HERE = os.environ.get('DDSP_GOLDEN_OUT') or HERE        # (tests/test_tf_backend_plumbing.py writes into a temp dir)
    B_, bname, bver = backend()
    tag = dict(backend=np.array(bname), backend_versions=np.array(bver))
    if os.path.exists(REF_CKPT):
        ir = dafx22_ir()
        np.savez_compressed(os.path.join(HERE, 'dafx22_reverb_ir.npz'), ir=ir, rows=np.array([0, 9]),
                            backend=np.array('reference-data'), backend_versions=np.array('dafx22/ckpt-0'))
    else:                      # a TF host without the checkpoint blob: the committed rows are the same data
        ir = np.load(os.path.join(committed, 'dafx22_reverb_ir.npz'))['ir']
    cases = recalled_cases(B_)
    np.savez_compressed(os.path.join(HERE, 'recalled_details.npz'), **cases, **tag)
    report_recalled(cases)

    # ---- C1: 1 s mono note, 64 harmonics, 24 kHz, dry -----------------------------------------
    rng = np.random.default_rng(1234)
    T, H, sr = 250, 64, 24000
    raw = synth_controls(rng, 1, T, H, S=1, silent_frac=0.0, midi_lo=45, midi_hi=45)
    synth = B_.MultiInharmonic(frame_rate=250, sample_rate=sr, inference=True)
    ctl = synth.get_controls(raw['amplitudes'], raw['harmonic_distribution'], raw['inharm_coef'], raw['f0_hz'])
    audio = synth.get_signal(**ctl)
    np.savez_compressed(os.path.join(HERE, 'c1_mono.npz'), sample_rate=sr, frame_rate=250,
                        **{f'raw_{k}': v for k, v in raw.items()}, **{f'ctl_{k}': v for k, v in ctl.items()},
                        audio=audio.astype(np.float32), **tag)

    # ---- down-sized C2: B=1, P=2, T=50, dafx22 dims, full chain --------------------------------
    rng = np.random.default_rng(1234)
    B, P, T, H, K, S, sr = 1, 2, 50, 96, 64, 2, 16000
    N = T * (sr // 250)
    feats = {}
    for i in range(P):
        for k, v in synth_controls(rng, B, T, H, S=S, K=K, silent_frac=0.0).items():
            feats[f'{k}_{i}'] = v
    feats['reverb_ir'] = np.ascontiguousarray(ir[:1, :6000])
    noises = np.stack([rng.uniform(-1, 1, [B, N]).astype(np.float32) for _ in range(P)], axis=0)
    additive = B_.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True)
    noise = B_.FilteredNoise(name='noise', frame_rate=250, sample_rate=sr)
    dag = B_.polyphonic_dag(additive, noise, B_.Reverb(name='reverb'),
                            additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'],
                            noise_controls=['magnitudes'], reverb_controls=['reverb_ir'], n_synths=P)
    out = B_.ProcessorGroup(dag)(feats, return_outputs_dict=True,
                                 extra_kwargs={'noise': [{'noise': z} for z in noises]})
    np.savez_compressed(os.path.join(HERE, 'c2_small.npz'), sample_rate=sr, frame_rate=250, n_synths=P,
                        noises=noises, audio=out['signal'].astype(np.float32),
                        dry=out['controls']['add']['signal'].astype(np.float32),
                        **{f'in_{k}': v for k, v in feats.items()}, **tag)
    # ---- SurrogateAdditive (configs/surrogate.gin: exp_tanh, no normalisation): one voice, 16 kHz, decaying partials ---
    rng = np.random.default_rng(4321)
    T, H, sr = 60, 96, 16000
    raw = synth_controls(rng, 1, T, H, S=1, silent_frac=0.0, midi_lo=50, midi_hi=50)
    decays = rng.uniform(0.9985, 1.0004, [1, T, H]).astype(np.float32)
    decay_time = (np.arange(T, dtype=np.float32) % 25)[None, :, None]
    sur = B_.SurrogateAdditive(frame_rate=250, sample_rate=sr, inference=True, scale_fn=B_.exp_tanh,
                               normalize_harm_distribution=False)
    sctl = sur.get_controls(raw['amplitudes'], decays, decay_time, raw['harmonic_distribution'], raw['inharm_coef'], raw['f0_hz'])
    saudio = sur.get_signal(**sctl)
    np.savez_compressed(os.path.join(HERE, 'c3_surrogate.npz'), sample_rate=sr, frame_rate=250, raw_decays=decays,
                        raw_decay_time=decay_time, **{f'raw_{k}': v for k, v in raw.items()},
                        **{f'ctl_{k}': np.asarray(v, np.float32) for k, v in sctl.items()}, audio=saudio.astype(np.float32), **tag)
    # ---- FeedbackDelayNetwork.get_ir: two rooms drawn like sub_modules.py:386-418, one damped, one lively --
    rng = np.random.default_rng(777)
    D, E, sr = 8, 200, 16000
    rooms = dict(input_gain=rng.normal(0.25, 0.1, [2, D]), output_gain=rng.normal(0.25, 0.1, [2, D]),
                 gain_allpass=rng.normal(0.25, 0.1, [2, D, 4]), delays_allpass=rng.normal(400.0, 60.0, [2, D, 4]),
                 time_rev_0_sec=np.asarray([0.4, 1.8]), alpha_tone=1.0 / (1.0 + np.exp(-rng.normal(0.0, 0.1, [2]))),
                 early_ir=rng.normal(0.0, 0.1, [2, E]))
    rooms = {k: np.asarray(v, np.float32) for k, v in rooms.items()}
    irs = np.stack([np.asarray(B_.fdn_get_ir(*[rooms[k][i] for k in ('input_gain', 'output_gain', 'gain_allpass',
                                                                       'delays_allpass', 'time_rev_0_sec', 'alpha_tone',
                                                                       'early_ir')], sampling_rate=float(sr)), np.float32)
                    for i in range(2)])
    np.savez_compressed(os.path.join(HERE, 'c4_fdn_ir.npz'), sample_rate=sr, ir=irs, **{f'p_{k}': v for k, v in rooms.items()}, **tag)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')


if __name__ == '__main__':
    main()
