"""CPU: the committed golden vectors, their provenance, and the oracle against them.

Every fixture made by tests/golden/make_golden.py carries a `backend` field: "restatement" (expected outputs
computed by oracle/ddsp_oracle.py -- all that is possible without TensorFlow) or "tf" (the same inputs through the
real TensorFlow + ddsp 3.7.0 + ddsp_piano modules, tests/golden/tf_backend.py).  The day the generator has run on a
TF host these tests pin the ORACLE to reference outputs (and, per recalled ddsp detail, say which switch is wrong if
one is); the -m gpu golden tests pin the HIP path to the same files.  A half-upgraded fixture set is an error."""
import glob
import os

import numpy as np
import pytest

from util import O, rms, rms_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SYNTH_FIXTURES = ('c1_mono', 'c2_small', 'c3_surrogate', 'c4_fdn_ir', 'recalled_details')
KEYS = dict(additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'],
            noise_controls=['magnitudes'], reverb_controls=['reverb_ir'])


def _load(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def _backends():
    return {n: str(_load(n)['backend']) for n in SYNTH_FIXTURES}


def test_every_fixture_names_its_backend_and_the_set_is_consistent():
    for path in glob.glob(os.path.join(GOLD, '*.npz')):
        g = np.load(path)
        if os.path.basename(path) == 'midi_conditioning.npz':
            continue                                   # outputs of the reference class itself (make_golden_midi.py)
        assert 'backend' in g.files, f'{path} has no backend field: regenerate with tests/golden/make_golden.py'
    kinds = set(_backends().values())
    assert kinds <= {'restatement', 'tf'}
    assert len(kinds) == 1, (f'mixed fixture set {_backends()}: a TF golden exists next to restatement goldens -- '
                             'rerun DDSP_GOLDEN_BACKEND=tf python tests/golden/make_golden.py for all of them')


def _tol():
    # restatement goldens ARE the oracle's outputs (bit for bit up to numpy's libm); TF goldens are the parity bar
    return 1e-4 if set(_backends().values()) == {'tf'} else 1e-6


def test_oracle_reproduces_config1():
    g = _load('c1_mono')
    syn = O.MultiInharmonic(frame_rate=int(g['frame_rate']), sample_rate=int(g['sample_rate']), inference=True)
    ctl = syn.get_controls(g['raw_amplitudes'], g['raw_harmonic_distribution'], g['raw_inharm_coef'], g['raw_f0_hz'])
    for k in ('amplitudes', 'harmonic_distribution', 'harmonic_shifts', 'f0_hz'):
        np.testing.assert_allclose(ctl[k], g[f'ctl_{k}'], rtol=2e-5, atol=1e-8)
    audio = syn.get_signal(**ctl)
    assert rms_err(audio, g['audio']) < _tol(), 'oracle vs golden C1: see test_recalled_details_match for the culprit'


def test_oracle_reproduces_the_surrogate_voice():
    g = _load('c3_surrogate')
    syn = O.SurrogateAdditive(frame_rate=int(g['frame_rate']), sample_rate=int(g['sample_rate']), inference=True,
                              scale_fn=O.exp_tanh, normalize_harm_distribution=False)
    ctl = syn.get_controls(g['raw_amplitudes'], g['raw_decays'], g['raw_decay_time'], g['raw_harmonic_distribution'],
                           g['raw_inharm_coef'], g['raw_f0_hz'])
    for k in ('amplitudes', 'decays', 'harmonic_distribution', 'harmonic_shifts'):
        np.testing.assert_allclose(ctl[k], g[f'ctl_{k}'], rtol=2e-5, atol=1e-8)
    assert rms_err(syn.get_signal(**ctl), g['audio']) < _tol()


def test_oracle_reproduces_the_fdn_impulse_responses():
    """FeedbackDelayNetwork.get_ir for a damped and a lively room.  Restatement goldens are the oracle's
    complex64-faithful mode; TF-made ones measure how far TensorFlow's complex64 inverse is from it (the lively room is
    ill-conditioned near its resonances: the float64 solve is what both approximate)."""
    g = _load('c4_fdn_ir')
    for i in range(2):
        args = [g[f'p_{k}'][i] for k in ('input_gain', 'output_gain', 'gain_allpass', 'delays_allpass', 'time_rev_0_sec',
                                         'alpha_tone', 'early_ir')]
        got = O.fdn_get_ir(*args, sampling_rate=float(g['sample_rate']))
        exact = O.fdn_get_ir(*args, sampling_rate=float(g['sample_rate']), exact_solve=True)
        tol = 1e-6 if _tol() < 1e-5 else 5e-3
        assert rms_err(got, g['ir'][i]) < tol * rms(g['ir'][i]), i
        assert rms_err(exact, g['ir'][i]) < 5e-3 * rms(g['ir'][i]), i


def test_oracle_reproduces_small_config2():
    g = _load('c2_small')
    P, sr = int(g['n_synths']), int(g['sample_rate'])
    feats = {k[3:]: g[k] for k in g.files if k.startswith('in_')}
    additive = O.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True)
    noise = O.FilteredNoise(name='noise', frame_rate=250, sample_rate=sr)
    dag = O.polyphonic_dag(additive, noise, O.Reverb(name='reverb'), n_synths=P, **KEYS)
    out = O.ProcessorGroup(dag)(feats, return_outputs_dict=True,
                                extra_kwargs={'noise': [{'noise': z} for z in g['noises']]})
    assert rms_err(out['controls']['add']['signal'], g['dry']) < _tol()
    assert rms_err(out['signal'], g['audio']) < _tol() * max(1.0, rms(g['audio']))


@pytest.mark.parametrize('detail', ['auto_delay', 'window_crop', 'resize', 'angular_cumsum', 'angular_offsets', 'angular_wrap',
                                    'upsamplers_bitwise', 'exp_sigmoid',
                                    'initial_bias', 'framed_fft_convolve', 'reverb_dry_mask', 'window_end_points',
                                    'exp_tanh', 'multi_add_order'])
def test_recalled_details_match(detail):
    """One single-operator case per recalled ddsp detail: with TF goldens a failure here names the detail whose
    default in oracle.RECALLED (and ddsp_piano_amd.core.RECALLED) is the wrong recollection."""
    g = _load('recalled_details')
    tol = 1e-5 if _tol() > 1e-5 else 1e-6
    if detail == 'auto_delay':
        got, want = O.frequency_filter(g['noise'], np.ones([1, 10, 96], np.float32), 257), g['flat_full']
    elif detail == 'window_crop':
        got, want = O.frequency_filter(g['noise'], np.ones([1, 10, 200], np.float32), 257), g['flat_crop']
    elif detail == 'resize':
        got, want = O.resample(g['ramp'], 12 * 96), g['ramp_linear']
        assert rms_err(O.resample(g['ramp'], 12 * 96, method='window'), g['ramp_window']) < tol
    elif detail == 'angular_cumsum':
        got, want = O.angular_cumsum(g['omega']), g['phase']
        d = np.abs(np.angle(np.exp(1j * (got.astype(np.float64) - want.astype(np.float64)))))
        assert d.max() < 1e-3                     # float32 scans of 2500 terms agree to ~1e-4 rad, never to 0.01
        return
    elif detail == 'angular_offsets':
        # 301 chunks, omega near pi / near 0 / in between (make_golden.py): float32 scans are deterministic, so the oracle
        # under its defaults reproduces a restatement golden BIT FOR BIT; a TF golden within a few ulp of the phase (a
        # different scan order inside tf.cumsum) still separates the variants, which lie 4e-5 rad rms apart
        ph = O.angular_cumsum(g['omega_long'])
        got = np.concatenate([ph[:, ::41].ravel(), ph[:, -1000:].ravel()])
        want = np.concatenate([g['phase_long_strided'].ravel(), g['phase_long_tail'].ravel()])
        if _tol() < 1e-5:
            assert np.array_equal(got, want)
        d = np.abs(np.angle(np.exp(1j * (got.astype(np.float64) - want.astype(np.float64)))))
        with O.recalled(angular_offsets='plain'):
            other = O.angular_cumsum(g['omega_long'])
        other = np.concatenate([other[:, ::41].ravel(), other[:, -1000:].ravel()])
        d_other = np.abs(np.angle(np.exp(1j * (other.astype(np.float64) - want.astype(np.float64)))))
        assert np.sqrt(np.mean(d ** 2)) < 0.2 * np.sqrt(np.mean(d_other ** 2)), \
            "the golden phases are closer to angular_offsets='plain' than to the default 'wrapped'"
        assert np.sqrt(np.mean(d ** 2)) < 1e-5
        return
    elif detail == 'angular_wrap':
        # the eighth switch (round 6): does angular_cumsum wrap what it returns?  The golden holds the function's OUTPUT for 301
        # chunks of a partial near Nyquist: wrapped it lies in [0, 2 pi), unwrapped it reaches ~3000 rad -- the two settings are
        # 1e3 rad apart, whatever the backend's rounding
        want = np.concatenate([g['phase_long_strided'].ravel(), g['phase_long_tail'].ravel()])
        wrapped = bool(want.max() < 6.2831855)
        with O.recalled(angular_wrap='final' if wrapped else 'none'):
            ph = O.angular_cumsum(g['omega_long'])
        got = np.concatenate([ph[:, ::41].ravel(), ph[:, -1000:].ravel()])
        assert rms_err(got, want) < 1e-3
        assert wrapped == (O.RECALLED_DEFAULTS['angular_wrap'] == 'final'), \
            "the golden phases say angular_cumsum's final `% 2 pi` is " + ('present' if wrapped else 'absent') + \
            ": oracle.RECALLED_DEFAULTS['angular_wrap'] is the wrong recollection"
        with O.recalled(angular_wrap='none'):
            unwrapped = O.angular_cumsum(g['omega_long'])
        assert unwrapped.max() > 1000.0 and np.abs(np.mod(unwrapped[:, :1000], O.TWO_PI_F32) - ph[:, :1000]).max() < 1e-3 or not wrapped
        return
    elif detail == 'upsamplers_bitwise':
        for key, val in (('rs_linear_96', O.resample(g['rs_in'], 37 * 96)), ('rs_linear_nonint', O.resample(g['rs_in'], 1000)),
                         ('rs_window_96', O.resample(g['rs_in'], 37 * 96, method='window'))):
            if _tol() < 1e-5 or key.startswith('rs_linear'):
                # the bilinear resize is three float32 operations per value with no freedom of order: bitwise even vs TF
                assert np.array_equal(val, g[key]), key
            assert rms_err(val, g[key]) < tol, key
        return
    elif detail == 'exp_sigmoid':
        got, want = O.exp_sigmoid(g['x']), g['exp_sigmoid']
    elif detail == 'framed_fft_convolve':          # n_samples % n_frames != 0: frame = ceil(N / F), the last frame padded
        got, want = O.fft_convolve(g['fc_audio'], g['fc_ir'], 'same', -1), g['fc_same']
        assert got.shape == want.shape == (1, 1000)
        v = O.fft_convolve(g['fc_audio'], g['fc_ir'], 'valid', 0)
        assert v.shape == g['fc_valid_delay0'].shape and rms_err(v, g['fc_valid_delay0']) < tol
    elif detail == 'reverb_dry_mask':
        got, want = O.Reverb(add_dry=True).get_signal(g['rv_audio'], g['rv_ir']), g['rv_wet_dry']
        wet = O.Reverb(add_dry=False).get_signal(g['rv_audio'], g['rv_ir'])
        assert rms_err(wet, g['rv_wet']) < tol and rms_err(got - wet, g['rv_audio']) < 1e-6      # ir[:, 0] = 5 is masked
    elif detail == 'window_end_points':
        got, want = O.resample(g['up_in'], 5 * 32, method='window'), g['up_window']
    elif detail == 'exp_tanh':
        got, want = O.exp_tanh(g['x']), g['exp_tanh']
    elif detail == 'multi_add_order':
        got, want = O.multi_add(list(g['ma_in'])), g['ma_sum']
        assert np.array_equal(got, want) or _tol() > 1e-5           # ((s0 + s1) + s2) + s3 in float32: order matters at 3e7
    else:
        got, want = O.FilteredNoise().get_controls(g['raw_mag'])['magnitudes'], g['noise_controls']
    assert rms_err(got, want) < tol, f'recalled detail {detail!r}: oracle default disagrees with the {_backends()} golden'
