"""GPU parity against the committed golden fixtures (tests/golden/*.npz, made by make_golden.py),
through the C-ABI, at BASELINE config 1 and the down-sized config 2; plus size-independent
properties at the full BASELINE sizes where the numpy oracle would take minutes."""
import os

import numpy as np
import pytest
import torch

from util import rms, rms_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-4          # BASELINE.json north_star: audio within 1e-4 RMS (float32) of the reference


def _dev(x):
    return torch.as_tensor(np.asarray(x), device='cuda')


def test_config1_mono_note():
    import ddsp_piano_amd as dp
    g = np.load(os.path.join(GOLD, 'c1_mono.npz'))
    syn = dp.MultiInharmonic(frame_rate=int(g['frame_rate']), sample_rate=int(g['sample_rate']), inference=True)
    ctl = syn.get_controls(_dev(g['raw_amplitudes']), _dev(g['raw_harmonic_distribution']),
                           _dev(g['raw_inharm_coef']), _dev(g['raw_f0_hz']))
    for k in ('harmonic_shifts', 'f0_hz'):
        assert np.array_equal(ctl[k].cpu().numpy(), g[f'ctl_{k}'])
    for k in ('amplitudes', 'harmonic_distribution'):
        np.testing.assert_allclose(ctl[k].cpu().numpy(), g[f'ctl_{k}'], rtol=2e-5, atol=1e-9)
    audio = syn.get_signal(**{k: _dev(g[f'ctl_{k}']) for k in
                              ('amplitudes', 'harmonic_distribution', 'harmonic_shifts', 'f0_hz')}).cpu().numpy()
    err = rms_err(audio, g['audio'])
    assert audio.shape == (1, 24000) and err < TOL, f'{err:.3e} (signal rms {rms(g["audio"]):.3e})'
    # end to end from the raw controls (get_controls on the GPU too)
    audio2 = syn(_dev(g['raw_amplitudes']), _dev(g['raw_harmonic_distribution']), _dev(g['raw_inharm_coef']),
                 _dev(g['raw_f0_hz'])).cpu().numpy()
    assert rms_err(audio2, g['audio']) < TOL


def test_surrogate_voice():
    """configs/surrogate.gin's synthesiser: get_controls (two kernels) and get_signal (the fused decay kernel) against the
    golden voice; with TF-made goldens this pins SURVEY.md row f-3 too."""
    import ddsp_piano_amd as dp
    g = np.load(os.path.join(GOLD, 'c3_surrogate.npz'))
    syn = dp.SurrogateAdditive(frame_rate=int(g['frame_rate']), sample_rate=int(g['sample_rate']), inference=True,
                               scale_fn=dp.exp_tanh, normalize_harm_distribution=False)
    ctl = syn.get_controls(_dev(g['raw_amplitudes']), _dev(g['raw_decays']), _dev(g['raw_decay_time']),
                           _dev(g['raw_harmonic_distribution']), _dev(g['raw_inharm_coef']), _dev(g['raw_f0_hz']))
    for k in ('harmonic_shifts', 'decays'):
        assert np.array_equal(ctl[k].cpu().numpy(), g[f'ctl_{k}']), k
    for k in ('amplitudes', 'harmonic_distribution'):
        np.testing.assert_allclose(ctl[k].cpu().numpy(), g[f'ctl_{k}'], rtol=2e-5, atol=1e-9)
    audio = syn.get_signal(**ctl).cpu().numpy()
    assert audio.shape == g['audio'].shape and rms_err(audio, g['audio']) < TOL


def test_fdn_impulse_responses():
    """FeedbackDelayNetwork.get_ir on the GPU (csrc/fdn.hip, float64 solve and the complex64-inverse switch) against the golden
    rooms: 5e-4 for the damped room, 5e-3 for the lively one (complex64-inverse goldens are that far from the exact solve)."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    g = np.load(os.path.join(GOLD, 'c4_fdn_ir.npz'))
    sr = int(g['sample_rate'])
    args = [_dev(g[f'p_{k}']) for k in ('input_gain', 'output_gain', 'gain_allpass', 'delays_allpass', 'time_rev_0_sec',
                                        'alpha_tone', 'early_ir')]
    for mode in ('float64', 'complex64'):
        prev = core.set_recalled(fdn_solve=mode)
        try:
            ir = dp.fdn_impulse_response(*args, sampling_rate=float(sr)).cpu().numpy()
        finally:
            core.set_recalled(**prev)
        assert ir.shape == g['ir'].shape
        for i, tol in enumerate((5e-4, 5e-3)):
            assert rms_err(ir[i], g['ir'][i]) < tol * rms(g['ir'][i]), (mode, i, rms_err(ir[i], g['ir'][i]) / rms(g['ir'][i]))


def test_config2_small_full_chain_with_real_dafx22_ir():
    import ddsp_piano_amd as dp
    g = np.load(os.path.join(GOLD, 'c2_small.npz'))
    P, sr = int(g['n_synths']), int(g['sample_rate'])
    feats = {k[3:]: _dev(g[k]) for k in g.files if k.startswith('in_')}
    additive = dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True)
    noise = dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr)
    dag = dp.polyphonic_dag(additive, noise, dp.Reverb(name='reverb'),
                            additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'],
                            noise_controls=['magnitudes'], reverb_controls=['reverb_ir'], n_synths=P)
    out = dp.ProcessorGroup(dag)(feats, return_outputs_dict=True, noise=[_dev(z) for z in g['noises']])
    err = rms_err(out['signal'].cpu().numpy(), g['audio'])
    assert err < TOL * max(1.0, rms(g['audio'])), f'{err:.3e} vs rms {rms(g["audio"]):.3e}'
    assert rms_err(out['controls']['add']['signal'].cpu().numpy(), g['dry']) < TOL


def test_recalled_detail_cases():
    """The single-operator cases that decide the recalled ddsp details (tests/golden/recalled_details.npz), through the
    HIP path.  With TF-made goldens this is where a wrong default of ddsp_piano_amd.core.RECALLED shows, by name."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    g = np.load(os.path.join(GOLD, 'recalled_details.npz'))
    noise = _dev(g['noise'])
    flat = core.frequency_filter(noise, torch.ones(1, 10, 96, device='cuda'), 257).cpu().numpy()
    assert rms_err(flat, g['flat_full']) < 1e-5, 'auto_delay'
    crop = core.frequency_filter(noise, torch.ones(1, 10, 200, device='cuda'), 257).cpu().numpy()
    assert rms_err(crop, g['flat_crop']) < 1e-5, 'window_crop / auto_delay'
    assert rms_err(core.resample(_dev(g['ramp']), 12 * 96).cpu().numpy(), g['ramp_linear']) < 1e-6, 'resize'
    assert rms_err(core.resample(_dev(g['ramp']), 12 * 96, method='window').cpu().numpy(), g['ramp_window']) < 1e-6
    np.testing.assert_allclose(core.exp_sigmoid(_dev(g['x'])).cpu().numpy(), g['exp_sigmoid'], rtol=2e-5, atol=1e-9)
    ctl = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=24000).get_controls(_dev(g['raw_mag']))['magnitudes']
    np.testing.assert_allclose(ctl.cpu().numpy(), g['noise_controls'], rtol=2e-5, atol=1e-9)
    # angular cumsum of a constant omega = a single sinusoid of constant frequency through the oscillator bank
    om = float(g['omega'][0, 0, 0])
    fe = torch.full((1, 2500 + 4, 1), om * 24000.0 / 6.2831855, device='cuda')          # omega = fe * 2 pi / sr
    got = core.cos_oscillator_bank(fe, torch.ones_like(fe), 24000, True, True).cpu().numpy()[0, :2500]
    assert np.abs(got - np.cos(g['phase'][0, :, 0].astype(np.float64))).max() < 2e-3, 'angular_cumsum'
    # ... and the case that separates the variants of angular_cumsum (301 chunks, omega near pi / near 0 / in between,
    # make_golden.py): the oscillator bank driven with the frequencies whose omegas the golden phases were scanned from
    fe = torch.repeat_interleave(_dev(g['fe_long_blocks']), 250, dim=0)[None].contiguous()         # [1, 301000, 3]
    cosines = core.cos_oscillator_bank(fe, torch.ones_like(fe), 24000, False, True).cpu().numpy()
    n = fe.shape[1]
    assert np.abs(cosines[:, ::41] - np.cos(g['phase_long_strided'].astype(np.float64))).max() < 3e-6, 'angular_offsets / angular_cumsum'
    assert np.abs(cosines[:, n - 1000:] - np.cos(g['phase_long_tail'].astype(np.float64))).max() < 3e-6, 'angular_offsets / angular_cumsum'
    # bitwise where bitwise is possible: the legacy bilinear resize (three float32 operations per value, one order)
    rs = _dev(g['rs_in'])
    assert np.array_equal(core.resample(rs, 37 * 96).cpu().numpy(), g['rs_linear_96']), 'resize, bitwise'
    assert np.array_equal(core.resample(rs, 1000).cpu().numpy(), g['rs_linear_nonint']), 'resize (N % T != 0), bitwise'
    up = core.resample(rs, 37 * 96, method='window').cpu().numpy()
    assert np.abs(up - g['rs_window_96']).max() <= 2.4e-7 * np.abs(g['rs_window_96']).max(), 'upsample_with_windows (the cross-fade is one FMA: 1 ulp)'


def test_full_size_properties_config3_shape():
    """At 3 s x poly 16 x H=128 (one batch row of config 3) the oracle is too slow for a unit test;
    check properties that do not depend on size instead."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    torch.manual_seed(0)
    R, T, H, sr = 16, 750, 128, 24000
    N = T * 96
    f0 = torch.full((R, T, 1), 0.0, device='cuda')
    f0[:, :, 0] = torch.linspace(30.0, 3000.0, R, device='cuda')[:, None]
    amp = torch.rand(R, T, 1, device='cuda')
    hd = torch.rand(R, T, H, device='cuda')
    inh = torch.full((R, T, 1), 3e-4, device='cuda')
    syn = dp.MultiInharmonic(sample_rate=sr, inference=True, scale_fn=None)
    ctl = syn.get_controls(amp, hd, inh, f0)
    y = syn.get_signal(**ctl)
    assert y.shape == (R, N) and torch.isfinite(y).all()
    # (a) linear in the amplitudes (phases do not depend on them): power-of-two scaling is bit exact
    ctl2 = dict(ctl)
    ctl2['amplitudes'] = ctl['amplitudes'] * 0.5
    assert torch.equal(syn.get_signal(**ctl2), y * 0.5)
    # (b) rows are independent: any sub-batch gives the same rows, bit for bit
    sub = {k: v[3:7].contiguous() for k, v in ctl.items()}
    assert torch.equal(syn.get_signal(**sub), y[3:7])
    # (c) the time-span decomposition never changes a bit
    a = core.harmonic_synthesis_fused(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T), ctl['harmonic_distribution'],
                                      ctl['harmonic_shifts'], N, sr, True, spans=1)
    b = core.harmonic_synthesis_fused(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T), ctl['harmonic_distribution'],
                                      ctl['harmonic_shifts'], N, sr, True, spans=72)
    assert torch.equal(a, b)
    # (d) bounded by the amplitude envelope: |y| <= sum_k a_k (normalised distribution -> amp)
    assert (y.abs().max(dim=1).values <= ctl['amplitudes'].reshape(R, T).max(dim=1).values * 1.0001 + 1e-6).all()
    # (e) reverb with a delta at tap d is a delay-and-add, at the full 3 s + 3 s IR FFT size (2^18)
    ir = torch.zeros(R, 72000, device='cuda')
    ir[:, 12345] = 0.5
    wet = dp.Reverb().get_signal(y, ir)
    exp = y.clone()
    exp[:, 12345:] += 0.5 * y[:, :-12345]
    assert (wet - exp).abs().max().item() < 2e-5 * max(1.0, y.abs().max().item())
