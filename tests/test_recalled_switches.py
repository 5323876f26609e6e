"""The recalled ddsp details as switches, product side (ddsp_piano_amd.core.RECALLED) against the oracle's
(oracle.ddsp_oracle.RECALLED): for every setting the library and the oracle agree, so whichever recollection a
TF + ddsp host confirms (tests/golden/make_golden.py, DDSP_GOLDEN_BACKEND=tf) is one assignment away.

CPU part: the host-built tables.  GPU part: the kernels driven by them and the C-ABI delay codes."""
import numpy as np
import pytest
import torch

from util import O, rms_err


@pytest.fixture
def core():
    from ddsp_piano_amd import core as c
    saved = dict(c.RECALLED)
    yield c
    c.RECALLED.update(saved)
    if saved.get('angular_offsets') is not None:          # (a library option as well: restore it there too)
        try:
            c.set_recalled(angular_offsets=saved['angular_offsets'])
        except RuntimeError:                              # no libddspp.so built: nothing to restore
            pass


@pytest.mark.parametrize('rule', ['legacy', 'half_pixel'])
def test_linear_tables_follow_the_resize_rule(core, rule):
    for T, N in ((750, 72000), (37, 1000), (12, 12 * 64)):
        lo, hi, w, aligned = core._linear_tables_np(T, N, rule)
        with O.recalled(resize=rule):
            olo, ohi, ow = O.linear_resample_positions(T, N)
        assert np.array_equal(lo, olo) and np.array_equal(hi, ohi) and np.array_equal(w, ow)
        if rule == 'half_pixel':
            assert not aligned                  # the fused oscillator path is refused, the operator route runs
    core.set_recalled(resize=rule)
    assert core.fused_synthesis_supported(750, 72000) == (rule == 'legacy')


@pytest.mark.parametrize('rule', ['ddsp370', 'centred'])
@pytest.mark.parametrize('K,ws', [(200, 257), (129, 257), (96, 257), (200, 101)])
def test_fir_matrix_follows_the_window_crop_rule(core, rule, K, ws):
    m = core._fir_matrix_np(K, ws, rule).astype(np.float64)
    rng = np.random.default_rng(K)
    mags = rng.uniform(0, 1, [3, K]).astype(np.float32)
    with O.recalled(window_crop=rule):
        ref = O.frequency_impulse_response(mags, ws)
    assert m.shape == (K, ref.shape[-1])
    assert np.abs(mags.astype(np.float64) @ m - ref).max() < 2e-6


def test_set_recalled_validates(core):
    with pytest.raises(KeyError):
        core.set_recalled(nope=1)
    with pytest.raises(ValueError):
        core.set_recalled(auto_delay='full')
    prev = core.set_recalled(auto_delay='half')
    assert prev['auto_delay'] == 'ddsp370' and core._auto_delay(-1) == -2 and core._auto_delay(7) == 7
    core.set_recalled(**prev)
    assert core._auto_delay(-1) == -1


def test_angular_offsets_variants_of_the_oracle():
    """angular_cumsum under both recollections of the offset sum (round 5): both are the exact phase mod 2 pi up to
    float32 round-off; they part ways from the third chunk on, by the rounding of `phase + offsets` at the magnitude of the sum."""
    rng = np.random.default_rng(9)
    n_chunks = 300
    om = np.repeat(rng.uniform(0.0, 3.1, [n_chunks * 4, 2]).astype(np.float32), 250, axis=0)[None]
    first = np.mod(np.cumsum(om[:, :1000], axis=1, dtype=np.float32), np.float32(2 * np.pi))
    with O.recalled(angular_offsets='plain'):
        a = O.angular_cumsum(om)
    b = O.angular_cumsum(om)
    for ph in (a, b):
        assert ph.shape == om.shape and (ph >= 0).all() and (ph < 6.2831855).all()
        assert np.array_equal(ph[:, :1000], first)                    # the first chunk has no offset
    # the second chunk's offset is one end phase (< 2 pi) under both; from then on the rounding of `phase + offsets` differs:
    # at the magnitude of the running sum (~ pi x chunks) under 'plain', at that of 2 pi under 'wrapped'
    assert np.array_equal(a[:, :2000], b[:, :2000]) and not np.array_equal(a[:, -1000:], b[:, -1000:])
    d = np.abs(np.angle(np.exp(1j * (a.astype(np.float64) - b.astype(np.float64)))))
    late = d[:, -50000:]
    assert late.max() > 0.5 * np.spacing(np.float32(np.pi * n_chunks)) and late.max() < 1e-3
    # (a chunk's own phase reaches 3100 rad at omega ~ pi, so `phase + offsets` is rounded at ulp(3100) = 2.4e-4 rad under
    # either rule; the rules differ in WHICH multiple of float32(2 pi) rides along, i.e. in that rounding -- ~5e-5 rad rms)
    assert 1e-6 < np.sqrt(np.mean(late ** 2)) < 2e-4
    # (both are float32 restatements of the same exact phase: ~0.07 rad away from it after 300 000 samples, SURVEY.md fact 8,
    # and 1e-4 rad from each other -- which recollection is right only shows against the real library)


# ------------------------------------------------------------------------------------------------ GPU
def _dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), device='cuda')


@pytest.mark.gpu
@pytest.mark.parametrize('rule', ['ddsp370', 'half'])
@pytest.mark.parametrize('K,U', [(96, 96), (64, 64), (200, 96), (65, 64)])
def test_frequency_filter_under_both_delay_rules(core, rule, K, U):
    """fused kernel (K = 96), two-call form (K = 64 at 16 kHz), generic FIR (K = 200: cropped window; K = 65)."""
    rng = np.random.default_rng(K + U)
    B, T = 2, 40
    noise = rng.uniform(-1, 1, [B, T * U]).astype(np.float32)
    mags = rng.uniform(0, 1, [B, T, K]).astype(np.float32)
    core.set_recalled(auto_delay=rule)
    got = core.frequency_filter(_dev(noise), _dev(mags), 257).cpu().numpy()
    with O.recalled(auto_delay=rule):
        ref = O.frequency_filter(noise, mags, 257)
    assert rms_err(got, ref) < 1e-5
    # and the known answer: a flat spectrum through the full-length window
    if K == 96:
        flat = core.frequency_filter(_dev(noise), torch.ones(B, T, K, device='cuda'), 257).cpu().numpy()
        d = 2 if rule == 'ddsp370' else 0
        assert np.abs(flat[:, d:] - noise[:, :T * U - d]).max() < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize('rule', ['ddsp370', 'half'])
def test_single_frame_fft_convolve_under_both_delay_rules(core, rule):
    rng = np.random.default_rng(3)
    audio = rng.normal(size=[2, 3000]).astype(np.float32)
    ir = rng.normal(size=[2, 501]).astype(np.float32)
    core.set_recalled(auto_delay=rule)
    got = core.fft_convolve(_dev(audio), _dev(ir)).cpu().numpy()            # delay_compensation = -1
    with O.recalled(auto_delay=rule):
        ref = O.fft_convolve(audio, ir)
    assert rms_err(got, ref) < 1e-5 * max(1.0, float(np.sqrt(np.mean(ref ** 2))))


@pytest.mark.gpu
@pytest.mark.parametrize('rule', ['ddsp370', 'centred'])
def test_cropped_window_design_under_both_crop_rules(core, rule):
    rng = np.random.default_rng(4)
    mags = rng.uniform(0, 1, [2, 9, 200]).astype(np.float32)
    core.set_recalled(window_crop=rule)
    got = core.frequency_impulse_response(_dev(mags), 257).cpu().numpy()
    with O.recalled(window_crop=rule):
        ref = O.frequency_impulse_response(mags, 257)
    assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize('rule', ['legacy', 'half_pixel'])
def test_harmonic_synthesis_under_both_resize_rules(core, rule):
    """'half_pixel' is refused by the fused kernel and takes resample -> cos_oscillator_bank; both match the oracle."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(5)
    B, T, H, sr = 2, 30, 64, 24000
    f0 = np.broadcast_to(rng.uniform(100, 900, [B, 1, 1]), [B, T, 1]).astype(np.float32).copy()
    f0 *= (1 + 0.01 * rng.normal(size=[B, T, 1])).astype(np.float32)
    amp = rng.normal(-1, 1, [B, T, 1]).astype(np.float32)
    hd = rng.normal(0, 1, [B, T, H]).astype(np.float32)
    inh = np.full([B, T, 1], 2e-4, np.float32)
    core.set_recalled(resize=rule)
    g = dp.MultiInharmonic(sample_rate=sr, inference=True)
    got = g(_dev(amp), _dev(hd), _dev(inh), _dev(f0)).cpu().numpy()
    with O.recalled(resize=rule):
        ref = O.MultiInharmonic(sample_rate=sr, inference=True)(amp, hd, inh, f0)
        x = rng.normal(size=[B, T, 5]).astype(np.float32)
        assert np.array_equal(core.resample(_dev(x), T * 96).cpu().numpy(), O.resample(x, T * 96))
    assert rms_err(got, ref) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('rule', ['wrapped', 'plain'])
def test_oscillator_kernels_under_both_offset_rules(core, rule):
    """ddsp.core.angular_cumsum's offset sum, wrapped or not (round 5, the seventh switch): the materialised kernel, the
    fused per-voice kernel, the compacted bank (the mix and every voice's stem) and a streamed render against the oracle under the same rule -- and, so that
    the case decides something, NOT against the oracle under the other rule.  340 chunks, partials up to Nyquist."""
    import ddsp_piano_amd as dp
    other = 'plain' if rule == 'wrapped' else 'wrapped'
    rng = np.random.default_rng(21)
    sr, U = 24000, 96
    # (1) cos_oscillator_bank on materialised envelopes, unsummed: cos(phase) per oscillator
    N, H = 340000, 4
    fe = np.repeat(rng.uniform(20.0, 11990.0, [N // 250, H]).astype(np.float32), 250, axis=0)[None]
    fe[..., 0] = np.float32(11950.0)                                   # omega ~ 3.128 all along
    ae = np.ones_like(fe)
    core.set_recalled(angular_offsets=rule)
    got = core.cos_oscillator_bank(_dev(fe), _dev(ae), sr, False, True).cpu().numpy()
    with O.recalled(angular_offsets=rule):
        ref = O.cos_oscillator_bank(fe, ae, sr, sum_sinusoids=False, use_angular_cumsum=True)
    with O.recalled(angular_offsets=other):
        ref_other = O.cos_oscillator_bank(fe, ae, sr, sum_sinusoids=False, use_angular_cumsum=True)
    assert np.abs(got - ref).max() < 3e-6, np.abs(got - ref).max()
    assert np.abs(got - ref_other)[:, -100000:].max() > 5e-5                   # the variants are 1e-4 apart late in the signal
    # (2) fused per-voice kernel, (3) compacted bank: a long poly-3 segment, high notes (partials up to Nyquist)
    B, P, T, Hh = 1, 3, 3542, 64                                               # 340 032 samples = 341 chunks
    from util import synth_controls
    voices = [synth_controls(rng, B, T, Hh, S=1, K=8, silent_frac=0.0, midi_lo=80, midi_hi=100) for _ in range(P)]
    syn = dp.MultiInharmonic(sample_rate=sr, inference=True)
    osyn = O.MultiInharmonic(sample_rate=sr, inference=True)
    keys = ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')
    with O.recalled(angular_offsets=rule):
        refs = [osyn(*[v[k] for k in keys]) for v in voices]
    with O.recalled(angular_offsets=other):
        ref0_other = osyn(*[voices[0][k] for k in keys])
    g0 = syn(*[_dev(voices[0][k]) for k in keys]).cpu().numpy()
    assert rms_err(g0, refs[0]) < 1e-5 * max(1.0, float(np.sqrt(np.mean(refs[0] ** 2))))
    assert rms_err(g0, ref0_other) > 10 * rms_err(g0, refs[0])
    stacked = {k: _dev(np.concatenate([v[k] for v in voices], axis=0)) for k in keys}        # rows [B * P] with B = 1
    ctl = syn._controls(stacked['amplitudes'], stacked['harmonic_distribution'], stacked['inharm_coef'], stacked['f0_hz'],
                        want_counts=True, want_shifts=False)
    mix = core.polyphonic_additive(ctl['f0_hz'], ctl['amplitudes'].reshape(P, T), ctl['harmonic_distribution'], None, B, T * U,
                                   sr, voice_major=False, audible=ctl['_audible'],
                                   inharm_coef=ctl['_inharm_coef'].reshape(P, T)).cpu().numpy()
    want = (refs[0] + refs[1]) + refs[2]
    assert rms_err(mix, want) < 1e-5 * max(1.0, float(np.sqrt(np.mean(want ** 2))))
    # (4) every voice's stem from the same packing (ddspp_polyphonic_stems)
    stems = core.polyphonic_stems(ctl['f0_hz'], ctl['amplitudes'].reshape(P, T), ctl['harmonic_distribution'], None, B, T * U, sr,
                                  audible=ctl['_audible'], inharm_coef=ctl['_inharm_coef'].reshape(P, T)).cpu().numpy()
    for i in range(P):
        assert rms_err(stems[i:i + 1], refs[i]) < 1e-5 * max(1.0, float(np.sqrt(np.mean(refs[i] ** 2)))), i
    assert rms_err(stems[0:1], ref0_other) > 10 * rms_err(stems[0:1], refs[0])
