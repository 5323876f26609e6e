"""GPU parity: oscillator bank, upsamplers and harmonic_synthesis against the CPU oracle.

Tolerance: BASELINE.json north_star -- audio within 1e-4 RMS (float32) of the reference on identical
inputs.  The phase arithmetic is bit-faithful, so the observed error is float32 round-off of cos and
of the harmonic sum (~1e-7); tests assert 1e-5 RMS to leave no room for an order-of-operations slip.
"""
import numpy as np
import pytest
import torch

from util import O, rms, rms_err, set_option, synth_controls

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev(x):
    return torch.as_tensor(x, device='cuda')


def _envelopes(rng, B, N, H, sr, fmax_scale=1.3):
    """Slowly varying positive frequency envelopes (some above Nyquist) and amplitudes."""
    f0 = rng.uniform(30.0, sr / 2 * fmax_scale / H, [B, 1, 1])
    k = np.arange(1, H + 1)[None, None, :]
    vib = 1.0 + 0.01 * np.sin(2 * np.pi * 5.0 * np.arange(N)[None, :, None] / sr + rng.uniform(0, 6, [B, 1, 1]))
    fe = (f0 * k * vib).astype(np.float32)
    ae = (rng.uniform(0, 1, [B, N, H]) / H).astype(np.float32)
    return fe, ae


@pytest.mark.parametrize('B,N,H,sr,angular,spans', [
    (2, 4000, 64, 16000, True, 1),
    (2, 4000, 64, 16000, True, 4),
    (1, 24000, 64, 24000, True, 0),       # BASELINE config 1 shape: 1 s, 64 harmonics
    (3, 7000, 96, 24000, True, 3),        # partial last chunk, VPL=2 with idle lanes
    (2, 5000, 128, 24000, True, 2),
    (1, 3000, 192, 32000, True, 0),       # VPL=3
    (2, 2400, 48, 8000, True, 0),
    (2, 4000, 64, 16000, False, 0),       # plain tf.cumsum (training path), short
    (1, 2048, 7, 22050, True, 0),         # odd sinusoid count
])
def test_cos_oscillator_bank_matches_oracle(B, N, H, sr, angular, spans):
    from ddsp_piano_amd import core
    rng = np.random.default_rng(1234 + N + H)
    fe, ae = _envelopes(rng, B, N, H, sr)
    ref = O.cos_oscillator_bank(fe, ae, sample_rate=sr, use_angular_cumsum=angular)
    got = core.cos_oscillator_bank(_dev(fe), _dev(ae), sample_rate=sr, use_angular_cumsum=angular,
                                   spans=spans).cpu().numpy()
    assert got.shape == ref.shape
    err = rms_err(got, ref)
    assert err < TOL, f'rms err {err:.3e} (signal rms {rms(ref):.3e})'


def test_spans_are_bitwise_equivalent():
    """Cutting time into spans (pre-pass + offset scan) must not change a single bit of the phase."""
    from ddsp_piano_amd import core
    rng = np.random.default_rng(7)
    fe, ae = _envelopes(rng, 2, 12000, 96, 24000)
    a = core.cos_oscillator_bank(_dev(fe), _dev(ae), 24000, use_angular_cumsum=True, spans=1)
    b = core.cos_oscillator_bank(_dev(fe), _dev(ae), 24000, use_angular_cumsum=True, spans=12)
    c = core.cos_oscillator_bank(_dev(fe), _dev(ae), 24000, use_angular_cumsum=True, spans=5)
    assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize('B,N,H,sr,spans', [
    (3, 5000, 64, 16000, 1),          # one wavefront per row, a partial last tile (5000 = 156 x 32 + 8)
    (2, 7000, 128, 24000, 1),         # the graded shape's two wavefronts per row
    (2, 7000, 128, 24000, 3),         # ... started from the span offsets of the pre-pass
    (1, 3008, 192, 32000, 2),
    (1, 2016, 256, 48000, 1),         # four wavefronts per row
    (2, 8, 64, 16000, 1),             # a single block: nothing but the ring's tail
    (1, 1040, 128, 24000, 1),         # one chunk boundary, then 40 samples
])
def test_stream_kernel_writes_what_osc_kernel_writes(monkeypatch, B, N, H, sr, spans):
    """osc_stream.hip (round 6: the HBM-bound shape's own kernel) against oscillator.hip's osc_kernel on the same envelopes:
    same phases, same order of the harmonic sum -> the same audio bit for bit; frequencies at 0, above Nyquist and (one
    column) negative so that the fast and the generic block forms both run."""
    from ddsp_piano_amd import core
    rng = np.random.default_rng(99 + N + H)
    fe, ae = _envelopes(rng, B, N, H, sr)
    fe[:, N // 3:N // 2, 5] = 0.0
    fe[0, N // 2:, 7] = -40.0
    fe[0, : N // 4, 9] = 1e-33
    set_option(monkeypatch, 'DDSPP_OSC_STREAM', 1)
    new = core.cos_oscillator_bank(_dev(fe), _dev(ae), sr, use_angular_cumsum=True, spans=spans)
    set_option(monkeypatch, 'DDSPP_OSC_STREAM', 0)
    old = core.cos_oscillator_bank(_dev(fe), _dev(ae), sr, use_angular_cumsum=True, spans=spans)
    assert torch.equal(new, old)
    ref = O.cos_oscillator_bank(fe, ae, sample_rate=sr, use_angular_cumsum=True)
    assert rms_err(new.cpu().numpy(), ref) < TOL


def test_unsummed_sinusoids():
    from ddsp_piano_amd import core
    rng = np.random.default_rng(11)
    fe, ae = _envelopes(rng, 1, 3000, 64, 16000)
    ref = O.cos_oscillator_bank(fe, ae, 16000, sum_sinusoids=False, use_angular_cumsum=True)
    got = core.cos_oscillator_bank(_dev(fe), _dev(ae), 16000, sum_sinusoids=False,
                                   use_angular_cumsum=True).cpu().numpy()
    assert got.shape == ref.shape == (1, 3000, 64)
    assert np.abs(got - ref).max() < 2e-6


def test_above_nyquist_partials_contribute_exactly_zero():
    from ddsp_piano_amd import core
    N, H, sr = 2000, 64, 16000
    fe = np.full([1, N, H], 9000.0, np.float32)       # >= sr / 2 everywhere
    fe[..., 0] = 8000.0                                # exactly Nyquist is removed too (>=)
    ae = np.ones([1, N, H], np.float32)
    got = core.cos_oscillator_bank(_dev(fe), _dev(ae), sr, use_angular_cumsum=True).cpu().numpy()
    assert np.all(got == 0.0)


def test_negative_and_huge_frequencies_take_the_generic_path():
    from ddsp_piano_amd import core
    rng = np.random.default_rng(5)
    N, H, sr = 3000, 64, 16000
    fe = rng.uniform(-3000.0, 3000.0, [1, N, H]).astype(np.float32)
    fe[0, :, 3] = 3.0e9                                 # phase beyond the fast-mod range
    fe[0, :, 5] = 1e-33                                 # denormal-quotient territory for the division
    ae = (rng.uniform(0, 1, [1, N, H]) / H).astype(np.float32)
    ref = O.cos_oscillator_bank(fe, ae, sr, use_angular_cumsum=True)
    got = core.cos_oscillator_bank(_dev(fe), _dev(ae), sr, use_angular_cumsum=True).cpu().numpy()
    assert rms_err(got, ref) < TOL


@pytest.mark.parametrize('T,U,C', [(50, 96, 128), (40, 64, 96), (25, 32, 48), (30, 96, 1), (20, 100, 6)])
def test_resample_kernels_are_bit_exact(T, U, C):
    from ddsp_piano_amd import core
    rng = np.random.default_rng(T * U + C)
    x = rng.normal(0, 1, [2, T, C]).astype(np.float32)
    N = T * U
    lin = core.resample(_dev(x), N).cpu().numpy()
    win = core.resample(_dev(x), N, method='window').cpu().numpy()
    assert np.array_equal(lin, O.resample(x, N))
    assert np.array_equal(win, O.resample(x, N, method='window'))


def test_resample_linear_non_integer_ratio():
    from ddsp_piano_amd import core
    rng = np.random.default_rng(3)
    x = rng.normal(0, 1, [2, 37, 8]).astype(np.float32)
    got = core.resample(_dev(x), 1000).cpu().numpy()
    assert np.array_equal(got, O.resample(x, 1000))


def test_resample_errors_mirror_ddsp():
    from ddsp_piano_amd import core
    x = torch.zeros(1, 10, 4, device='cuda')
    with pytest.raises(ValueError):
        core.resample(x, 95, method='window')        # n_timesteps % n_intervals != 0
    with pytest.raises(ValueError):
        core.resample(x, 10, method='window')        # more frames than time steps
    with pytest.raises(ValueError):
        core.resample(x, 100, method='bogus')
    with pytest.raises(ValueError):
        core.upsample_with_windows(torch.zeros(10, 4, device='cuda'), 100)


@pytest.mark.parametrize('B,T,H,S,sr,fr,angular', [
    (2, 60, 128, 1, 24000, 250, True),    # maestro-v2 dims
    (2, 60, 96, 2, 16000, 250, True),     # dafx22 dims: two sub-strings
    (1, 250, 64, 1, 24000, 250, True),    # BASELINE config 1: 1 s mono, 64 harmonics
    (2, 40, 48, 1, 8000, 250, True),
    (1, 30, 192, 1, 32000, 250, True),
    (2, 20, 96, 2, 16000, 250, False),    # training path (plain cumsum), short
])
def test_multi_inharmonic_get_signal_matches_oracle(B, T, H, S, sr, fr, angular):
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(99 + T + H)
    raw = synth_controls(rng, B, T, H, S=S, silent_frac=0.0)
    osynth = O.MultiInharmonic(frame_rate=fr, sample_rate=sr, inference=angular)
    gsynth = dp.MultiInharmonic(frame_rate=fr, sample_rate=sr, inference=angular)
    ctl = osynth.get_controls(raw['amplitudes'], raw['harmonic_distribution'], raw['inharm_coef'], raw['f0_hz'])
    ref = osynth.get_signal(**ctl)
    got = gsynth.get_signal(**{k: _dev(v) for k, v in ctl.items()}).cpu().numpy()
    assert got.shape == ref.shape == (B, T * (sr // fr))
    err = rms_err(got, ref)
    assert err < TOL, f'rms err {err:.3e} (signal rms {rms(ref):.3e})'


def test_fused_route_equals_three_operator_route_bitwise_phase():
    """harmonic_synthesis fused from frame controls vs resample -> resample -> cos_oscillator_bank."""
    from ddsp_piano_amd import core
    rng = np.random.default_rng(21)
    B, T, H, U, sr = 2, 40, 96, 96, 24000
    raw = synth_controls(rng, B, T, H, silent_frac=0.0)
    ctl = O.InHarmonic(sample_rate=sr, inference=True).get_controls(
        raw['amplitudes'], raw['harmonic_distribution'], raw['inharm_coef'], raw['f0_hz'])
    f0, amp, hd, sh = (_dev(ctl[k]) for k in ('f0_hz', 'amplitudes', 'harmonic_distribution', 'harmonic_shifts'))
    fused = core.harmonic_synthesis(f0, amp, sh, hd, n_samples=T * U, sample_rate=sr, use_angular_cumsum=True)
    hf = core.get_harmonic_frequencies(f0, H) * (1.0 + sh)
    fe = core.resample(hf, T * U)
    ae = core.resample(amp * hd, T * U, method='window')
    three = core.cos_oscillator_bank(fe, ae, sr, use_angular_cumsum=True)
    # identical phases; amplitudes differ by at most one fused rounding
    assert (fused - three).abs().max().item() < 1e-6


@pytest.mark.parametrize('scale,na,nb', [('exp_sigmoid', True, True), ('exp_tanh', False, True),
                                         (None, True, False), ('exp_sigmoid', False, False)])
def test_inharmonic_get_controls(scale, na, nb):
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(4)
    B, T, H, S, sr = 3, 25, 128, 2, 24000
    raw = synth_controls(rng, B, T, H, S=S)
    if scale is None:
        raw['amplitudes'] = np.abs(raw['amplitudes'])
        raw['harmonic_distribution'] = np.abs(raw['harmonic_distribution'])
    ofn = {'exp_sigmoid': O.exp_sigmoid, 'exp_tanh': O.exp_tanh, None: None}[scale]
    gfn = {'exp_sigmoid': dp.exp_sigmoid, 'exp_tanh': dp.exp_tanh, None: None}[scale]
    kw = dict(sample_rate=sr, normalize_after_nyquist_cut=na, normalize_below_nyquist=nb)
    ref = O.MultiInharmonic(scale_fn=ofn, **kw).get_controls(**raw)
    got = dp.MultiInharmonic(scale_fn=gfn, **kw).get_controls(**{k: _dev(v) for k, v in raw.items()})
    assert set(got) == set(ref)
    assert np.array_equal(got['harmonic_shifts'].cpu().numpy(), ref['harmonic_shifts'])   # exact ops only
    assert np.array_equal(got['f0_hz'].cpu().numpy(), ref['f0_hz'])
    for k in ('amplitudes', 'harmonic_distribution'):
        np.testing.assert_allclose(got[k].cpu().numpy(), ref[k], rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize('H,T,B,S', [(1, 1, 1, 1), (15, 3, 2, 1), (16, 9, 5, 2), (17, 7, 3, 1), (48, 26, 4, 1), (96, 13, 6, 2),
                                      (100, 5, 2, 1), (129, 11, 3, 1), (200, 6, 2, 2), (300, 4, 3, 1), (512, 2, 2, 1)])
def test_inharmonic_get_controls_shapes_counts_and_last_voice(H, T, B, S):
    """The get_controls kernel (one DPP row of 16 lanes per frame, harmonics in steps of 16, groups of 16 above Nyquist
    skipped whole) over harmonic counts that are not multiples of 16, rows shorter than a wavefront's eight frames,
    frame counts that leave the last wavefront half empty -- every flag combination against the oracle; the per-frame
    audible counts against the definition (1 + the last harmonic whose amp * hd is not zero); and the group form:
    harmonic_shifts of every segment's last voice only, both row layouts."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(H * 31 + T)
    sr = 16000
    raw = synth_controls(rng, B, T, H, S=S, midi_lo=30, midi_hi=108)
    dev = {k: _dev(v) for k, v in raw.items()}
    order = ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')
    for na in (True, False):
        for nb in (True, False):
            kw = dict(sample_rate=sr, normalize_after_nyquist_cut=na, normalize_below_nyquist=nb)
            ref = O.MultiInharmonic(scale_fn=O.exp_sigmoid, **kw).get_controls(**raw)
            syn = dp.MultiInharmonic(scale_fn=dp.exp_sigmoid, **kw)
            got = syn._controls(*[dev[k] for k in order], want_counts=True)
            assert np.array_equal(got['harmonic_shifts'].cpu().numpy(), ref['harmonic_shifts'])
            hd, amp = got['harmonic_distribution'].cpu().numpy(), got['amplitudes'].cpu().numpy()
            np.testing.assert_allclose(hd, ref['harmonic_distribution'], rtol=2e-5, atol=1e-9)
            np.testing.assert_allclose(amp, ref['amplitudes'], rtol=2e-5, atol=1e-9)
            assert ((ref['harmonic_distribution'] == 0) == (hd == 0)).all()              # the cut is the reference's cut
            live = (amp * hd) != 0
            want = np.where(live.any(-1), H - np.argmax(live[..., ::-1], axis=-1), 0)
            assert np.array_equal(got['_audible'].cpu().numpy() & 0xffff, want), (na, nb)
            for P in (1, B):                            # rows = [B / P segments, P voices] (or voice major)
                for vm in (False, True):
                    grp = syn._controls(*[dev[k] for k in order], want_counts=True, want_shifts=False, last_voice_of=(P, vm))
                    assert torch.equal(grp['harmonic_distribution'], got['harmonic_distribution'])
                    assert torch.equal(grp['_audible'], got['_audible'])
                    rows = np.arange(B).reshape(P, B // P)[-1] if vm else np.arange(B).reshape(B // P, P)[:, -1]
                    assert np.array_equal(grp['_shifts_last'].cpu().numpy(), ref['harmonic_shifts'][rows]), (P, vm)


@pytest.mark.parametrize('H,scale', [(128, 'exp_sigmoid'), (96, 'exp_tanh'), (64, None), (128, 'exp_tanh'), (48, 'exp_sigmoid'), (192, 'exp_sigmoid')])
def test_lean_get_controls_kernel_equals_the_generic_one_bitwise(H, scale, monkeypatch):
    """The default flags at 48 / 64 / 96 / 128 / 192 harmonics run inharmonic_controls_lean_kernel (controls.hip): the cut above
    Nyquist decided from the hardware square root, the correctly rounded one only for a wavefront with a partial within
    1e-6 of Nyquist.  Every output equals the all-purpose kernel's (DDSPP_CONTROLS_GENERIC=1) bit for bit -- also when
    partials are parked ON Nyquist (f0 = Nyquist / k exactly, with and without inharmonicity), in silent frames
    (f0 = 0) and in the half-empty last wavefront -- and the cut is the oracle's cut."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(H)
    B, T, S, sr = 6, 37, 1, 24000
    raw = synth_controls(rng, B, T, H, S=S, midi_lo=21, midi_hi=108)
    nyq = np.float32(sr / 2)
    f0 = raw['f0_hz']
    for i, k in enumerate((1, 2, 3, 7, 16, 17, 33, 64, H - 1, H)):
        f0[1, i, 0] = nyq / np.float32(k)                 # partial k lands on (or an ulp off) Nyquist
        f0[2, i, 0] = nyq / np.float32(k)
        raw['inharm_coef'][2, i] = 0.0                     # ... exactly on it
        f0[3, i, 0] = np.nextafter(nyq / np.float32(k), np.float32(0))
        raw['inharm_coef'][3, i] = 0.0
    f0[4, :5, 0] = 0.0
    if scale is None:
        raw['amplitudes'] = np.abs(raw['amplitudes'])
        raw['harmonic_distribution'] = np.abs(raw['harmonic_distribution'])
    ofn = {'exp_sigmoid': O.exp_sigmoid, 'exp_tanh': O.exp_tanh, None: None}[scale]
    gfn = {'exp_sigmoid': dp.exp_sigmoid, 'exp_tanh': dp.exp_tanh, None: None}[scale]
    dev = {k: _dev(v) for k, v in raw.items()}
    order = ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')
    syn = dp.MultiInharmonic(scale_fn=gfn, sample_rate=sr)
    runs = {}
    for generic in (1, 0):
        set_option(monkeypatch, 'DDSPP_CONTROLS_GENERIC', generic)
        full = syn._controls(*[dev[k] for k in order], want_counts=True)
        grp = syn._controls(*[dev[k] for k in order], want_counts=True, want_shifts=False, last_voice_of=(3, False))
        runs[generic] = {**{k: v.clone() for k, v in full.items() if torch.is_tensor(v)},
                         '_shifts_last': grp['_shifts_last'].clone(), 'hd_grp': grp['harmonic_distribution'].clone()}
    assert set(runs[0]) == set(runs[1])
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k
    ref = O.MultiInharmonic(scale_fn=ofn, sample_rate=sr).get_controls(**raw)
    hd = runs[0]['harmonic_distribution'].cpu().numpy()
    assert ((ref['harmonic_distribution'] == 0) == (hd == 0)).all()
    assert np.array_equal(runs[0]['harmonic_shifts'].cpu().numpy(), ref['harmonic_shifts'])
    np.testing.assert_allclose(hd, ref['harmonic_distribution'], rtol=2e-5, atol=1e-9)


def test_custom_python_scale_fn():
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(8)
    raw = synth_controls(rng, 2, 10, 64)
    ref = O.InHarmonic(scale_fn=lambda x: np.exp(np.minimum(x, 3.0)).astype(np.float32)).get_controls(**raw)
    got = dp.InHarmonic(scale_fn=lambda x: torch.exp(torch.clamp(x, max=3.0))).get_controls(
        **{k: _dev(v) for k, v in raw.items()})
    np.testing.assert_allclose(got['harmonic_distribution'].cpu().numpy(), ref['harmonic_distribution'],
                               rtol=2e-5, atol=1e-9)


def test_surrogate_additive_matches_oracle():
    """SurrogateAdditive (configs/surrogate.gin): decaying-amplitude oscillator bank."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(17)
    B, T, H, sr = 2, 40, 96, 16000
    raw = synth_controls(rng, B, T, H, silent_frac=0.0)
    decays = rng.uniform(0.9990, 1.0002, [B, T, H]).astype(np.float32)
    decay_time = np.tile((np.arange(T, dtype=np.float32) % 20)[None, :, None], [B, 1, 1])
    o = O.SurrogateAdditive(sample_rate=sr, scale_fn=O.exp_tanh, normalize_harm_distribution=False, inference=True)
    g = dp.SurrogateAdditive(sample_rate=sr, scale_fn=dp.exp_tanh, normalize_harm_distribution=False, inference=True)
    octl = o.get_controls(raw['amplitudes'], decays, decay_time, raw['harmonic_distribution'], raw['inharm_coef'], raw['f0_hz'])
    gctl = g.get_controls(_dev(raw['amplitudes']), _dev(decays), _dev(decay_time), _dev(raw['harmonic_distribution']),
                          _dev(raw['inharm_coef']), _dev(raw['f0_hz']))
    for k in ('amplitudes', 'decays', 'harmonic_distribution', 'harmonic_shifts'):
        np.testing.assert_allclose(gctl[k].cpu().numpy(), octl[k], rtol=2e-5, atol=1e-9)
    ref = o.get_signal(**octl)
    got = g.get_signal(**{k: _dev(v) for k, v in octl.items()}).cpu().numpy()
    assert got.shape == ref.shape == (B, T * 64)
    assert rms_err(got, ref) < TOL * max(1.0, rms(ref))


@pytest.mark.parametrize('B,T,H,sr,shifts', [
    (2, 40, 96, 16000, True),        # configs/surrogate.gin dims
    (3, 260, 96, 16000, True),       # 17 chunks: spans start inside frames (1000 % 64 != 0) -> the power is re-seeded mid-frame
    (2, 60, 130, 24000, False),      # three oscillators per lane, no harmonic_shifts
    (1, 25, 40, 48000, True),        # hop 192
])
def test_surrogate_fused_route_matches_oracle_and_the_operator_route(B, T, H, sr, shifts, monkeypatch):
    """Round 4: SurrogateAdditive.get_signal straight from the frame controls (ddspp_surrogate_harmonic_synthesis: the decay
    term inside the oscillator kernel) against the oracle and against the three-operator route over materialised envelopes."""
    from ddsp_piano_amd import core
    rng = np.random.default_rng(1000 * B + T + H)
    U = sr // 250
    raw = synth_controls(rng, B, T, H, silent_frac=0.0, midi_lo=30, midi_hi=80)
    decays = rng.uniform(0.9985, 1.0, [B, T, H]).astype(np.float32)
    decays[:, :, ::7] *= -1.0                                            # |decays|
    decays[0, :, 3] = 1e-5                                               # the clip's lower end: underflows within a frame
    decay_time = np.tile((np.arange(T, dtype=np.float32) % 37)[None, :, None], [B, 1, 1])
    o = O.SurrogateAdditive(sample_rate=sr, scale_fn=O.exp_tanh, normalize_harm_distribution=False, inference=True)
    octl = o.get_controls(raw['amplitudes'], np.abs(decays), decay_time, raw['harmonic_distribution'], raw['inharm_coef'],
                          raw['f0_hz'])
    octl['decays'] = np.where(octl['decays'] == np.abs(decays), decays, octl['decays']).astype(np.float32)   # signs back in
    if not shifts:
        octl['harmonic_shifts'] = None
    ref = O.surrogate_harmonic_synthesis(octl['f0_hz'], octl['amplitudes'], octl['decays'], octl['decay_time'],
                                         octl['harmonic_shifts'], octl['harmonic_distribution'], upsampling=U, sample_rate=sr,
                                         use_angular_cumsum=True)
    args = dict(frequencies=_dev(octl['f0_hz']), amplitudes=_dev(octl['amplitudes']), decays=_dev(octl['decays']),
                decay_time=_dev(octl['decay_time']), harmonic_shifts=_dev(octl['harmonic_shifts']) if shifts else None,
                harmonic_distribution=_dev(octl['harmonic_distribution']), upsampling=U, sample_rate=sr,
                use_angular_cumsum=True)
    set_option(monkeypatch, 'DDSPP_SURROGATE_MATERIALISED', 1)
    slow = core.surrogate_harmonic_synthesis(**args).cpu().numpy()
    set_option(monkeypatch, 'DDSPP_SURROGATE_MATERIALISED')
    fast = core.surrogate_harmonic_synthesis(**args).cpu().numpy()
    assert fast.shape == ref.shape == (B, T * U)
    assert rms_err(slow, ref) < TOL * max(1.0, rms(ref))
    assert rms_err(fast, ref) < TOL * max(1.0, rms(ref)), rms_err(fast, ref)
    assert np.abs(fast - slow).max() < 2e-5 * max(1.0, np.abs(slow).max())
    # plain cumsum (training-time form) takes the same kernel
    args['use_angular_cumsum'] = False
    fast_p = core.surrogate_harmonic_synthesis(**args).cpu().numpy()
    set_option(monkeypatch, 'DDSPP_SURROGATE_MATERIALISED', 1)
    slow_p = core.surrogate_harmonic_synthesis(**args).cpu().numpy()
    set_option(monkeypatch, 'DDSPP_SURROGATE_MATERIALISED')
    assert np.abs(fast_p - slow_p).max() < 5e-4 * max(1.0, np.abs(slow_p).max())


@pytest.mark.parametrize('B,T,H,S,sr,fr', [
    (2, 30, 64, 1, 16000, 160),     # U = 100: not a multiple of 8 -> resample + cos_oscillator_bank route
    (1, 1, 32, 1, 16000, 250),      # a single control frame
    (3, 7, 1, 1, 8000, 250),        # a single harmonic
    (1, 25, 130, 1, 24000, 250),    # H not a multiple of 64 nor of a vector width (VPL = 3, strided lanes)
    (2, 25, 200, 2, 24000, 250),    # S * H = 400 virtual oscillators (VPL = 8)
])
def test_get_signal_edge_shapes(B, T, H, S, sr, fr):
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(B * 1000 + T + H)
    raw = synth_controls(rng, B, T, H, S=S, silent_frac=0.0, midi_lo=40, midi_hi=70)
    o = O.MultiInharmonic(frame_rate=fr, sample_rate=sr, inference=True)
    g = dp.MultiInharmonic(frame_rate=fr, sample_rate=sr, inference=True)
    ctl = o.get_controls(raw['amplitudes'], raw['harmonic_distribution'], raw['inharm_coef'], raw['f0_hz'])
    ref = o.get_signal(**ctl)
    got = g.get_signal(**{k: _dev(v) for k, v in ctl.items()}).cpu().numpy()
    assert got.shape == ref.shape == (B, T * int(sr / fr))
    assert rms_err(got, ref) < TOL * max(1.0, rms(ref)), rms_err(got, ref)


def test_materialised_bank_with_many_sinusoids_and_groups():
    from ddsp_piano_amd import core
    rng = np.random.default_rng(123)
    for H, spans in [(256, 1), (320, 2), (66, 1), (5, 3)]:
        fe, ae = _envelopes(rng, 2, 3000, H, 48000)
        ref = O.cos_oscillator_bank(fe, ae, 48000, use_angular_cumsum=True)
        got = core.cos_oscillator_bank(_dev(fe), _dev(ae), 48000, use_angular_cumsum=True, spans=spans).cpu().numpy()
        assert rms_err(got, ref) < TOL, (H, spans, rms_err(got, ref))
