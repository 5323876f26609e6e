"""NoiseBandNetSynth / FilterBank (filtered_noise_synth.py:51-317): host filter bank and noise bands
against the oracle restatement (CPU), the band-modulation kernel against the oracle's get_signal loop (GPU)."""
import math

import numpy as np
import pytest
import torch

from util import O, rms, rms_err


def _phase(n_band, noise_len, seed=3):
    return np.random.default_rng(seed).uniform(-math.pi, math.pi, [n_band, noise_len // 2 + 1]).astype(np.float32)


def test_filter_bank_and_noise_bands_match_the_oracle():
    from ddsp_piano_amd import noise_band_net as nbn
    for n_band, sr in ((8, 16000), (16, 24000)):
        fb = nbn.FilterBank(n_filters_linear=n_band // 2, n_filters_log=n_band // 2, sample_rate=sr)
        ref = O.nbn_filterbank(n_band, sr)
        assert len(fb.filters) == len(ref) == n_band
        for a, b in zip(fb.filters, ref):
            assert len(a) == len(b) and len(a) % 2 == 1 and np.abs(a - b).max() < 1e-12
        assert len(fb.band_centers) == n_band and np.all(np.diff(fb.band_centers) > 0)
        noise_len = nbn.get_next_power_of_2(fb.max_filter_len)
        ph = _phase(n_band, noise_len)
        bands, nl = nbn.get_noise_bands(fb, 16, True, ph)
        obands, onl = O.nbn_noise_bands(ref, 16, True, ph)
        assert nl == onl == noise_len and bands.shape == (noise_len, n_band) and obands.shape == (1, noise_len, n_band)
        assert np.abs(bands - obands[0]).max() < 1e-6 and abs(np.abs(bands).max() - 1.0) < 1e-6
        # loopable: a band's spectrum has no DC / Nyquist line, every band is band limited around its centre
        spec = np.abs(np.fft.rfft(bands[:, n_band // 2].astype(np.float64)))
        peak_hz = np.argmax(spec) * sr / noise_len
        assert fb.band_centers[n_band // 2 - 1] * 0.5 < peak_hz < fb.band_centers[n_band // 2 + 1] * 1.5
    # the branch on which the reference dies with a NameError (filtered_noise_synth.py:108-109): a linear bank
    lin = nbn.FilterBank.get_frequency_bands(6, 6, 20, 1, 16000)
    assert lin.shape == (5, 2) and np.array_equal(lin, O.nbn_frequency_bands(6, 6, 20, 1, 16000))
    with pytest.raises(AssertionError):
        nbn.NoiseBandNetSynth(min_noise_len=24)


@pytest.mark.gpu
@pytest.mark.parametrize('T', [20, 256, 300, 640])          # shorter than a chunk, exactly one, a stretched tail, many
def test_noise_band_synth_matches_oracle(T):
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(T)
    n_band, sr, U, B = 8, 16000, 64, 2
    filters = O.nbn_filterbank(n_band, sr)
    noise_len = 2 ** math.ceil(math.log2(max(len(h) for h in filters)))
    ph = _phase(n_band, noise_len)
    obands, _ = O.nbn_noise_bands(filters, 16, True, ph)
    raw = rng.normal(0, 1, [B, T, n_band]).astype(np.float32)
    syn = dp.NoiseBandNetSynth(upsampling=U, sample_rate=sr, phase_noise=ph)
    ctl = syn.get_controls(torch.as_tensor(raw, device='cuda'))
    np.testing.assert_allclose(ctl['amplitudes'].cpu().numpy(), O.exp_sigmoid(raw), rtol=2e-5, atol=1e-9)
    for shift in (0, 12345, noise_len - 1):
        got = syn.get_signal(torch.as_tensor(O.exp_sigmoid(raw), device='cuda'), shift=shift).cpu().numpy()
        ref = O.nbn_get_signal(O.exp_sigmoid(raw), obands, noise_len, U, shift)
        assert got.shape == ref.shape == (B, T * U)
        assert rms_err(got, ref) < 1e-6 * max(1.0, rms(ref)), (T, shift)
    assert syn.noise_len == noise_len and int(syn.noise_len / U) == 256
    y = syn(torch.as_tensor(raw, device='cuda'))             # the processor call: own roll, finite audio
    assert y.shape == (B, T * U) and torch.isfinite(y).all()
