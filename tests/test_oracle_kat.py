"""CPU: known-answer tests that pin the oracle.  The reference ships no golden
vectors and TF/ddsp cannot be imported, so the restatement is pinned by analytic identities, by
independent numpy/scipy evaluations, and by the committed restatement goldens (drift guard)."""
import os

import numpy as np
import pytest
import scipy.signal as ss

from util import O, rms_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_single_partial_is_fp32_accumulated_cosine():
    """(i) constant f0, one-hot distribution, amp 1: y[n] = cos(sum_{m<=n} omega) in float32, n < 1000."""
    T, U, sr, H = 10, 96, 24000, 8
    f0 = np.full([1, T, 1], 441.0, np.float32)
    hd = np.zeros([1, T, H], np.float32)
    hd[..., 2] = 1.0
    amp = np.ones([1, T, 1], np.float32)
    y = O.harmonic_synthesis(f0, amp, np.zeros_like(hd), hd, n_samples=T * U, sample_rate=sr,
                             use_angular_cumsum=True)
    om = np.float32(np.float32(np.float32(441.0 * 3) * np.float32(2 * np.pi)) / np.float32(sr))
    ph = np.float32(0)
    exp = []
    for _ in range(T * U):
        ph = np.float32(ph + om)
        exp.append(np.cos(np.float32(np.mod(ph, np.float32(2 * np.pi)))))
    w = O.hann_window(2 * U)
    assert y.shape == (1, T * U)
    np.testing.assert_allclose(y[0], np.asarray(exp, np.float32), atol=2e-7)
    assert y[0, 0] == np.cos(om)                     # inclusive cumsum: first sample is cos(omega), not 1
    assert abs(float(w[0])) < 1e-7 and abs(float(w[U]) - 1.0) < 1e-7


def test_upsamplers_partition_of_unity_and_ramp():
    """(ii) constant in = constant out for the Hann OLA; a ramp stays a ramp with last-frame hold."""
    T, U = 12, 64
    c = np.full([1, T, 3], 0.37, np.float32)
    assert np.abs(O.resample(c, T * U, method='window') - 0.37).max() < 1e-7
    ramp = np.arange(T, dtype=np.float32)[None, :, None]
    lin = O.resample(ramp, T * U)[0, :, 0]
    expect = np.minimum(np.arange(T * U) / U, T - 1)
    np.testing.assert_allclose(lin, expect, atol=1e-5)
    # closed form of the OLA: y[n] = x[t] w[U + r] + x[t + 1] w[r]
    x = np.random.default_rng(0).normal(size=[2, T, 5]).astype(np.float32)
    w = O.hann_window(2 * U)
    xe = np.concatenate([x, x[:, -1:]], 1)
    n = np.arange(T * U)
    closed = (xe[:, n // U] * w[U + n % U][None, :, None]).astype(np.float32) + \
        (xe[:, n // U + 1] * w[n % U][None, :, None]).astype(np.float32)
    assert np.array_equal(O.resample(x, T * U, method='window'), closed.astype(np.float32))
    with pytest.raises(ValueError):
        O.resample(x, T * U + 1, method='window')


def test_linear_resample_is_legacy_bilinear():
    """lo == n // U for the shipped ratios; weight = frac(float32(n) * float32(T / N))."""
    for T, U in [(750, 96), (750, 64), (750, 32), (500, 128), (375, 192), (34000, 96)]:
        lo, hi, w = O.linear_resample_positions(T, T * U)
        assert np.array_equal(lo, np.arange(T * U) // U)
        assert np.all(hi <= T - 1) and np.all((w >= 0) & (w < 1))


def test_nyquist_mask_and_renormalisation():
    """(iii) a partial at or above sr/2 contributes exactly 0 and the rest renormalise to 1."""
    syn = O.InHarmonic(sample_rate=16000, scale_fn=None)
    T, H = 4, 8
    f0 = np.full([1, T, 1], 1500.0, np.float32)       # partials 6.. are >= 8000 Hz (with inharmonicity)
    ctl = syn.get_controls(np.ones([1, T, 1], np.float32), np.ones([1, T, H], np.float32),
                           np.full([1, T, 1], 1e-3, np.float32), f0)
    hd = ctl['harmonic_distribution']
    assert np.all(hd[..., 5:] == 0.0) and np.allclose(hd.sum(-1), 1.0, atol=1e-6)
    quiet = syn.get_controls(np.ones([1, T, 1], np.float32), np.ones([1, T, H], np.float32),
                             np.zeros([1, T, 1], np.float32), np.full([1, T, 1], 8.18, np.float32))
    assert np.all(quiet['amplitudes'] == 0.0)           # f0 <= min_frequency gates the voice


def test_flat_magnitudes_delay_two_samples():
    """(iv) VERIFY item: even-length full-Hann FIR -> unit tap at K - 1, net delay of 2 samples."""
    rng = np.random.default_rng(0)
    for K in (32, 64, 96, 128):
        ir = O.frequency_impulse_response(np.ones([1, 1, K], np.float32), 257)
        assert ir.shape[-1] == 2 * (K - 1) and int(np.argmax(ir[0, 0])) == K - 1
        assert abs(ir[0, 0, K - 1] - 1.0) < 1e-6
    noise = rng.uniform(-1, 1, [1, 960]).astype(np.float32)
    y = O.frequency_filter(noise, np.ones([1, 10, 96], np.float32), 257)
    assert np.abs(y[0, 2:] - noise[0, :-2]).max() < 1e-6
    ir = O.frequency_impulse_response(np.ones([1, 1, 200], np.float32), 257)    # cropped branch
    assert ir.shape[-1] == 257 and int(np.argmax(ir[0, 0])) == 127


def test_frequency_filter_equals_per_frame_convolution():
    rng = np.random.default_rng(1)
    T, U, K = 7, 96, 96
    noise = rng.uniform(-1, 1, [1, T * U])
    mags = rng.uniform(0, 1, [1, T, K]).astype(np.float32)
    ir = O.frequency_impulse_response(mags, 257).astype(np.float64)
    z = np.zeros(T * U + ir.shape[-1] - 1)
    for t in range(T):
        z[t * U: t * U + U + ir.shape[-1] - 1] += np.convolve(noise[0, t * U:(t + 1) * U], ir[0, t])
    start = (ir.shape[-1] - 1) // 2 - 1
    ref = z[start:start + T * U]
    got = O.frequency_filter(noise.astype(np.float32), mags, 257)
    assert rms_err(got[0], ref) < 2e-7


def test_reverb_kats_and_scipy():
    """(v), (vi)."""
    rng = np.random.default_rng(2)
    audio = rng.normal(size=[2, 3000]).astype(np.float32)
    ir = np.zeros([2, 500], np.float32)
    ir[:, 41] = 1.0
    ir[:, 0] = 9.0
    y = O.Reverb().get_signal(audio, ir)
    exp = audio.copy()
    exp[:, 41:] += audio[:, :-41]
    assert np.abs(y - exp).max() < 2e-6
    ir = rng.normal(size=[2, 700]).astype(np.float32)
    wet = O.fft_convolve(audio, ir, padding='same', delay_compensation=0)
    ref = np.stack([ss.fftconvolve(audio[b].astype(np.float64), ir[b].astype(np.float64))[:3000] for b in range(2)])
    assert rms_err(wet, ref) < 1e-6 * np.sqrt(np.mean(ref ** 2)) * 10
    with pytest.raises(ValueError):
        O.Reverb().get_controls(audio)
    with pytest.raises(ValueError):
        O.fft_convolve(audio, np.zeros([3, 10], np.float32))


def test_equal_substrings_equal_single_string():
    """(vii) MultiInharmonic with S = 2 equal f0 == S = 1 (amp / S twice)."""
    rng = np.random.default_rng(3)
    from util import synth_controls
    raw = synth_controls(rng, 1, 20, 32, S=1, silent_frac=0.0)
    one = O.MultiInharmonic(sample_rate=16000, inference=True)
    a = one(raw['amplitudes'], raw['harmonic_distribution'], raw['inharm_coef'], raw['f0_hz'])
    f2 = np.concatenate([raw['f0_hz'], raw['f0_hz']], -1)
    b = one(raw['amplitudes'], raw['harmonic_distribution'], raw['inharm_coef'], f2)
    assert rms_err(a, b) < 1e-6


def test_angular_cumsum_against_float64_on_short_signals():
    """The chunked scan differs from exact arithmetic only by float32 round-off on a short, low phase."""
    rng = np.random.default_rng(4)
    om = rng.uniform(0, 0.05, [1, 2500, 4]).astype(np.float32)
    got = O.angular_cumsum(om)
    ref = np.mod(np.cumsum(om.astype(np.float64), axis=1), 2 * np.pi)
    d = np.abs(np.exp(1j * got) - np.exp(1j * ref)).max()
    assert got.shape == om.shape and d < 1e-4


def test_polyphonic_dag_shape():
    add = O.MultiInharmonic(name='additive')
    nz = O.FilteredNoise(name='noise')
    dag = O.polyphonic_dag(add, nz, O.Reverb(), additive_controls=['a', 'b'], noise_controls=['m'],
                           reverb_controls=['reverb_ir'], n_synths=3)
    assert len(dag) == 10
    assert dag[2][1] == ['noise/signal', 'additive/signal']
    assert dag[5][1] == ['add/signal', 'noise/signal', 'additive/signal']
    assert dag[-1][1] == ['add/signal', 'reverb_ir'] and dag[3][0] is add and dag[4][0] is nz


def test_goldens_have_not_drifted():
    """The committed restatement goldens (tests/golden/make_golden.py) still come out of the oracle."""
    g = np.load(os.path.join(GOLD, 'c1_mono.npz'))
    syn = O.MultiInharmonic(frame_rate=int(g['frame_rate']), sample_rate=int(g['sample_rate']), inference=True)
    ctl = syn.get_controls(g['raw_amplitudes'], g['raw_harmonic_distribution'], g['raw_inharm_coef'], g['raw_f0_hz'])
    for k in ctl:
        np.testing.assert_allclose(ctl[k], g[f'ctl_{k}'], rtol=1e-6, atol=1e-9)
    audio = syn.get_signal(**{k: g[f'ctl_{k}'] for k in ctl})
    assert audio.shape == (1, 24000) and rms_err(audio, g['audio']) < 1e-6
    ir = np.load(os.path.join(GOLD, 'dafx22_reverb_ir.npz'))['ir']
    assert ir.shape == (2, 24000) and abs(ir[0, 1] - 3.18) < 0.05


# ----------------------------------------------------------------------------------------------------
# The recalled ddsp details as switches (oracle.RECALLED): every alternative has its own known answer, so the
# day real ddsp outputs exist the matching setting is identified by data, not by recollection.
# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('rule,delay_full,shift_crop', [('ddsp370', 2, 0), ('half', 0, -1)])
def test_auto_delay_rule_known_answers(rule, delay_full, shift_crop):
    """Flat magnitudes = a unit tap at the FIR's zero-time index: tap K - 1 for the full-length window
    (2 (K - 1) <= 257), tap 127 for the 257-tap crop.  'ddsp370' crops (L - 1) // 2 - 1 samples: 2-sample
    delay / exact alignment.  'half' crops L // 2: exact alignment / one sample early."""
    rng = np.random.default_rng(0)
    noise = rng.uniform(-1, 1, [1, 960]).astype(np.float32)
    with O.recalled(auto_delay=rule):
        y = O.frequency_filter(noise, np.ones([1, 10, 96], np.float32), 257)
        yc = O.frequency_filter(noise, np.ones([1, 10, 200], np.float32), 257)
    d = delay_full
    assert np.abs(y[0, d:] - noise[0, :960 - d]).max() < 1e-6
    # cropped window: the unit tap sits one sample before the window's peak and carries the weight hann257[127]
    g = O.hann_window(257)[127]
    if shift_crop == 0:
        assert np.abs(yc[0] - g * noise[0]).max() < 2e-6
    else:
        assert np.abs(yc[0, :-1] - g * noise[0, 1:]).max() < 2e-6
    assert O.RECALLED['auto_delay'] == 'ddsp370'           # the context manager restores the default


@pytest.mark.parametrize('rule,peak', [('ddsp370', 127), ('centred', 128)])
def test_window_crop_known_answers(rule, peak):
    with O.recalled(window_crop=rule):
        ir = O.frequency_impulse_response(np.ones([1, 1, 200], np.float32), 257)
    assert ir.shape[-1] == 257 and int(np.argmax(ir[0, 0])) == peak
    if rule == 'centred':           # symmetric about the peak, which carries the full window weight
        assert abs(ir[0, 0, peak] - 1.0) < 1e-6
        assert np.abs(ir[0, 0, peak + 1:] - ir[0, 0, peak - 1::-1][:128]).max() < 1e-6


@pytest.mark.parametrize('rule', ['legacy', 'half_pixel'])
def test_resize_rule_known_answers(rule):
    """A ramp stays a ramp: legacy -> y[n] = n / U with the last frame held; half-pixel -> shifted by half an
    output pixel, (n + 0.5) / U - 0.5, clamped to the first / last frame at the ends."""
    T, U = 12, 96
    x = np.arange(T, dtype=np.float32)[None, :, None]
    with O.recalled(resize=rule):
        y = O.resample(x, T * U)[0, :, 0]
    n = np.arange(T * U)
    if rule == 'legacy':
        want = np.minimum(n / U, T - 1)
    else:
        want = np.clip((n + 0.5) / U - 0.5, 0, T - 1)
    assert np.abs(y - want).max() < 1e-5


@pytest.mark.parametrize('rule', ['ddsp370', 'exclusive'])
def test_angular_cumsum_rule_known_answers(rule):
    """Constant omega: phase[n] = (n + 1) omega (inclusive: the first sample is cos(omega)) or n omega (exclusive:
    the first sample is 1), wrapped; both continue across the 1000-sample chunk boundary."""
    om = np.float32(0.01)
    with O.recalled(angular_cumsum=rule):
        ph = O.angular_cumsum(np.full([1, 2500, 1], om, np.float32))[0, :, 0]
    k = np.arange(2500) + (1 if rule == 'ddsp370' else 0)
    want = np.mod(k.astype(np.float64) * float(om), 2 * np.pi)
    err = np.abs(np.angle(np.exp(1j * (ph - want))))
    assert err.max() < 5e-4 and (ph[0] == (om if rule == 'ddsp370' else 0.0))


def test_scale_constants_are_switchable():
    x = np.float32(0.3)
    base = O.exp_sigmoid(x)
    with O.recalled(exp_sigmoid=(10.0, 1.0, 1e-7)):
        assert abs(O.exp_sigmoid(x) - (base - 1e-7) / 2 - 1e-7) < 1e-6
    with O.recalled(initial_bias=0.0):
        assert O.FilteredNoise().initial_bias == 0.0
    assert O.FilteredNoise().initial_bias == -5.0
    with pytest.raises(ValueError):
        O.recalled(auto_delay='nope')
    with pytest.raises(KeyError):
        O.recalled(unknown=1)
