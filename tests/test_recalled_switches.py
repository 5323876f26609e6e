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
