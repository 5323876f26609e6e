"""GPU parity for the first "next" row: FDN impulse-response generation."""
import numpy as np
import pytest
import torch

from util import O, rms, rms_err

pytestmark = pytest.mark.gpu


def _params(rng, B, D=8, A=4):
    return dict(input_gain=rng.normal(0.25, 0.1, [B, D]).astype(np.float32),
                output_gain=rng.normal(0.25, 0.1, [B, D]).astype(np.float32),
                gain_allpass=rng.normal(0.25, 0.1, [B, D, A]).astype(np.float32),
                delays_allpass=(np.tile(O.FDN_DELAYS_ALLPASS[None, :D, :A], [B, 1, 1]) +
                                rng.normal(0, 20, [B, D, A])).astype(np.float32),
                time_rev_0_sec=np.abs(rng.normal(2.0, 0.5, [B])).astype(np.float32),
                alpha_tone=(1 / (1 + np.exp(-rng.normal(0, 0.1, [B])))).astype(np.float32),
                early_ir=rng.normal(0, 0.1, [B, 200]).astype(np.float32))


@pytest.mark.parametrize('sr', [16000.0, 24000.0])
def test_fdn_impulse_response_matches_oracle(sr):
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(int(sr))
    B = 3
    prm = _params(rng, B)
    ref = np.stack([O.fdn_get_ir(**{k: v[b] for k, v in prm.items()}, sampling_rate=sr) for b in range(B)])
    exact = np.stack([O.fdn_get_ir(**{k: v[b] for k, v in prm.items()}, sampling_rate=sr, exact_solve=True)
                      for b in range(B)])
    got = dp.fdn_impulse_response(**{k: torch.as_tensor(v, device='cuda') for k, v in prm.items()},
                                  sampling_rate=sr).cpu().numpy()
    assert got.shape == ref.shape == (B, int(2 * sr))
    # I - F D is nearly singular at the network's resonances (|H| peaks at > 1000 x its median), so ONE ulp of
    # difference in a float32 transfer value (sincosf / powf of numpy vs the GPU maths library -- or of TF) moves
    # the bins around a resonance by cond x 6e-8 ~ 5e-4.  That is the agreement any two correct implementations
    # of the reference's float32 recipe can reach on a lively room; a damped room (below) agrees to round-off.
    # Kernel and oracle both use THE float32 value of every transcendental (evaluated in double, rounded once), so
    # against the float64 solve of the same float32 transfer values the agreement is round-off; the reference's own
    # complex64 inverse (ref) sits cond x 6e-8 away from both.
    err = rms_err(got, exact)
    assert err < 5e-4 * rms(exact), f'{err:.3e} vs rms {rms(exact):.3e}'      # complex64 products / quotients still differ by an ulp
    err = rms_err(got, ref)
    assert err < 2e-3 * rms(ref), f'{err:.3e} vs rms {rms(ref):.3e}'
    assert np.abs(got[:, :200] - ref[:, :200]).max() < 2e-3 * np.abs(ref).max()


def test_fdn_damped_room_agrees_to_roundoff():
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(3)
    B, sr = 2, 16000.0
    prm = _params(rng, B)
    prm['time_rev_0_sec'] = np.asarray([0.25, 0.4], np.float32)          # short reverb: no sharp resonances
    exact = np.stack([O.fdn_get_ir(**{k: v[b] for k, v in prm.items()}, sampling_rate=sr, exact_solve=True)
                      for b in range(B)])
    ref = np.stack([O.fdn_get_ir(**{k: v[b] for k, v in prm.items()}, sampling_rate=sr) for b in range(B)])
    got = dp.fdn_impulse_response(**{k: torch.as_tensor(v, device='cuda') for k, v in prm.items()},
                                  sampling_rate=sr).cpu().numpy()
    assert rms_err(got, exact) < 2e-5 * rms(exact) and rms_err(got, ref) < 2e-5 * rms(ref)


def test_fdn_processor_and_fractional_delays():
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(5)
    prm = {k: v[0] for k, v in _params(rng, 1, D=6).items()}
    delays = [233.3, 311.0, 421.7, 461.2, 587.5, 613.9]
    fdn = dp.FeedbackDelayNetwork(sampling_rate=8000.0, delay_values=delays)
    assert len(fdn) == 6 and fdn.freq_points == 16000
    audio = rng.normal(0, 0.1, [2, 8000]).astype(np.float32)
    ctl = fdn.get_controls(torch.as_tensor(audio, device='cuda'), **{k: torch.as_tensor(v, device='cuda') for k, v in prm.items()})
    ref_ir = O.fdn_get_ir(**prm, delay_values=np.asarray(delays, np.float32), sampling_rate=8000.0, exact_solve=True)
    assert rms_err(ctl['ir'].cpu().numpy(), ref_ir) < 2e-3 * rms(ref_ir)
    out = fdn.get_signal(**ctl).cpu().numpy()
    ref = O.fdn_get_signal(audio, ctl['ir'].cpu().numpy())           # the apply step, on the same ir
    assert rms_err(out, ref) < 1e-5 * rms(ref)
    own = dp.FeedbackDelayNetwork(trainable=True, delay_lines=6, delay_trainable=True, sampling_rate=8000.0)
    assert len(own) == 6 and set(own.parameters()) == set(own.PARAMETER_NAMES)
    with pytest.raises(ValueError):
        fdn.parameters()                       # a non-trainable network holds none
