"""GPU: the one-call group driver of the library (ddspp_group_*, csrc/group.cpp) through ddsp_piano_amd.NativeGroup
against ProcessorGroup's batched Python route -- the same kernels with the same arguments -- and a small case against
the oracle."""
import numpy as np
import pytest
import torch

from test_gpu_fuzz import musical_controls
from util import oracle_segments

pytestmark = pytest.mark.gpu
KEYS = dict(additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'], noise_controls=['magnitudes'])


def _setup(seed, B, P, T, H, K, S, U, vm, L, flags=None):
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(seed)
    sr = 250 * U
    raw = musical_controls(rng, B * P, T, H, S, sr)
    raw['magnitudes'] = rng.normal(0.0, 1.5, [B * P, T, K]).astype(np.float32)
    feats, host = {}, {}
    for k, v in raw.items():
        whole = v.reshape(*((P, B) if vm else (B, P)), T, v.shape[-1])
        dev = torch.as_tensor(whole, device='cuda')
        for i in range(P):
            feats[f'{k}_{i}'] = dev[i] if vm else dev[:, i]
            host[f'{k}_{i}'] = whole[i] if vm else whole[:, i]
    rk = []
    if L:
        ir = (rng.normal(0.0, 1.0, [B, L]) * np.exp(-6.9 * np.arange(L) / L)[None, :] * 0.05).astype(np.float32)
        feats['reverb_ir'], host['reverb_ir'] = torch.as_tensor(ir, device='cuda'), ir
        rk = ['reverb_ir']
    flags = dict(flags or {})
    scale = getattr(dp, flags.pop('scale', 'exp_sigmoid'))

    def group():
        return dp.ProcessorGroup(dp.polyphonic_dag(
            dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True, scale_fn=scale, **flags),
            dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr, scale_fn=scale),
            dp.Reverb(name='reverb') if L else None, n_synths=P, reverb_controls=rk, **KEYS))
    noise = rng.uniform(-1, 1, [B, P, T * U]).astype(np.float32)
    return dp, group, feats, host, noise, sr


def _close(a, b, path=''):
    """Same kernels, same arguments -- except the Hann tables: the library's host builder evaluates the float32 cosine
    with libm, the Python layer with numpy, and the two differ by an ulp in a few entries (tests/test_cabi.py)."""
    assert a.shape == b.shape, path
    err = (a - b).abs().max().item()
    assert err <= 2e-6 * max(1.0, float(b.abs().max())), (path, err)


def _same(a, b, path=''):
    if isinstance(a, dict):
        assert set(a) == set(b), (path, set(a) ^ set(b))
        for k in a:
            if k != 'inputs':
                _same(a[k], b[k], f'{path}/{k}')
    elif torch.is_tensor(a):
        _close(a, b, path)


@pytest.mark.parametrize('seed,B,P,T,H,K,S,U,vm,L,flags', [
    (1, 3, 16, 125, 128, 96, 1, 96, False, 4000, None), (2, 2, 16, 60, 128, 96, 1, 96, True, 2500, None),
    (3, 2, 3, 90, 96, 64, 2, 64, False, 3000, None), (4, 5, 3, 50, 96, 64, 1, 64, True, 0, None),
    (5, 4, 1, 40, 64, 32, 1, 128, False, 1000, None), (6, 2, 6, 75, 48, 32, 1, 32, False, 2000,
                                                       dict(scale='exp_tanh', normalize_after_nyquist_cut=False)),
    (7, 17, 16, 33, 128, 96, 1, 96, True, 1500, None), (8, 2, 3, 3, 64, 32, 1, 128, False, 500, None),
    (9, 3, 2, 2, 96, 96, 1, 96, True, 300, None)])          # (three frames, two frames: the last voice's harmonic_shifts too)
def test_native_group_equals_the_python_route(seed, B, P, T, H, K, S, U, vm, L, flags, monkeypatch):
    from util import set_option
    dp, group, feats, _, noise, sr = _setup(seed, B, P, T, H, K, S, U, vm, L, flags)
    z = torch.as_tensor(noise, device='cuda')
    if seed % 2 == 1:          # small batches keep per-voice noise rows; odd seeds force the kernel's voice sums in both routes
        set_option(monkeypatch, 'DDSPP_VOICE_SUMS', 8)
    py, nat = group(), dp.NativeGroup(group(), feats)
    junk = [torch.full((B, T * U), float('nan'), device='cuda') for _ in range(12)]     # what torch.empty hands out next
    del junk
    _close(nat(feats, noise=z), py(feats, noise=z), 'audio only')
    _same(nat(feats, return_outputs_dict=True, noise=z), py(feats, return_outputs_dict=True, noise=z))
    # the library's own noise stream: the first call of a fresh pair draws the same numbers (same seed, same counter)
    py2, nat2 = group(), dp.NativeGroup(group(), feats)
    _close(nat2(feats), py2(feats), 'own noise')
    a, b = nat2(feats), nat2(feats)
    assert (a - b).abs().max().item() > 1e-5            # a new draw per call
    with pytest.raises(ValueError):
        nat(({k: (v[:, :T - 1] if k != 'reverb_ir' else v) for k, v in feats.items()}))


def test_native_group_against_the_oracle():
    dp, group, feats, host, noise, sr = _setup(11, 2, 4, 100, 128, 96, 1, 96, False, 3000)
    nat = dp.NativeGroup(group(), feats)
    got = nat(feats, return_outputs_dict=True, noise=torch.as_tensor(noise, device='cuda'))
    ref = oracle_segments(host, noise, 4, sr, [0, 1])
    for b in (0, 1):
        r = ref[b]
        for name, x in (('signal', got['signal'][b]), ('dry', got['controls']['add']['signal'][b]),
                        ('additive_last', got['controls']['additive']['signal'][b]),
                        ('noise_last', got['controls']['noise']['signal'][b])):
            want = np.asarray(r[name]).reshape(-1)
            err = np.sqrt(np.mean((x.cpu().numpy().astype(np.float64) - want) ** 2))
            assert err < 1e-4 * max(1e-3, np.sqrt(np.mean(want.astype(np.float64) ** 2))), (name, err)


def test_native_group_side_stream(monkeypatch):
    """The noise branch and the impulse-response transform on the group's own stream (what large batches get), forced on
    at a small size; several calls back to back re-use the workspace across the fork / join."""
    from util import set_option
    set_option(monkeypatch, 'DDSPP_SIDE_STREAM', 1)
    set_option(monkeypatch, 'DDSPP_SIDE_STREAM_MIN', 1)
    dp, group, feats, _, noise, sr = _setup(21, 4, 16, 125, 128, 96, 1, 96, False, 5000)
    z = torch.as_tensor(noise, device='cuda')
    nat = dp.NativeGroup(group(), feats)
    set_option(monkeypatch, 'DDSPP_SIDE_STREAM_MIN', None)
    set_option(monkeypatch, 'DDSPP_SIDE_STREAM', None)
    want = group()(feats, return_outputs_dict=True, noise=z)
    for _ in range(3):
        got = nat(feats, return_outputs_dict=True, noise=z)
    _same(got, want)
    _close(nat(feats, noise=z), want['signal'], 'audio only')


def test_native_group_32khz_and_captured():
    """K = 128 / U = 128 (the ENSTDkCl 32 kHz dimensions: FilteredNoise in the two-call form), and the driver inside a
    replayed HIP graph (ddspp_group_run only launches kernels)."""
    dp, group, feats, _, noise, sr = _setup(31, 2, 4, 64, 192, 128, 1, 128, False, 2000,
                                            dict(scale='exp_tanh', normalize_after_nyquist_cut=False))
    z = torch.as_tensor(noise, device='cuda')
    py, nat = group(), dp.NativeGroup(group(), feats)
    _same(nat(feats, return_outputs_dict=True, noise=z), py(feats, return_outputs_dict=True, noise=z))
    fast = dp.CapturedGroup(nat, feats, return_outputs_dict=True)
    got = fast(feats, noise=z)
    _close(got['signal'], py(feats, noise=z), 'graph replay')
    _close(fast(feats, noise=z)['controls']['add']['signal'], py(feats, return_outputs_dict=True, noise=z)['controls']['add']['signal'])


@pytest.mark.parametrize('sr,H,K,lines,apply_only', [(8000, 48, 32, 6, False), (16000, 96, 64, 8, False), (16000, 96, 64, 8, True)])
def test_native_group_with_a_feedback_delay_network_last(sr, H, K, lines, apply_only):
    """The ENSTDkCl DAG (configs/ENSTDkCl-8kHz.gin:85-104, ENSTDkCl-16kHz.gin): exp_tanh, no renormalisation after the
    Nyquist cut, and a FeedbackDelayNetwork that holds its parameters as the last node (reverb_controls = []) -- through
    the one-call driver (reverb_keep_dry_tap = 1, reverb_add_dry = 0) against the Python route; also the apply step alone
    with the impulse response as a control.  New parameters reach the driver on the next call."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(sr + lines)
    B, P, T, S = 3, 4, 50, 1
    U = sr // 250
    raw = musical_controls(rng, B * P, T, H, S, sr)
    raw['magnitudes'] = rng.normal(0.0, 1.5, [B * P, T, K]).astype(np.float32)
    feats = {}
    for k, v in raw.items():
        dev = torch.as_tensor(v.reshape(B, P, T, v.shape[-1]), device='cuda')
        for i in range(P):
            feats[f'{k}_{i}'] = dev[:, i]
    z = torch.as_tensor(rng.uniform(-1, 1, [B, P, T * U]).astype(np.float32), device='cuda')
    fdn = dp.FeedbackDelayNetwork(trainable=True, delay_trainable=True, delay_lines=lines, sampling_rate=sr, name='fdn', seed=5)
    rk = []
    if apply_only:
        feats['reverb_ir'] = fdn.get_controls(z[:, 0])['ir'].clone()
        rk = ['reverb_ir']

    def group():
        return dp.ProcessorGroup(dp.polyphonic_dag(
            dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True, scale_fn=dp.exp_tanh,
                               normalize_after_nyquist_cut=False),
            dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr, scale_fn=dp.exp_tanh),
            dp.FeedbackDelayNetworkApply(name='fdn') if apply_only else fdn, n_synths=P, reverb_controls=rk, **KEYS))

    py, nat = group(), dp.NativeGroup(group(), feats)
    want = py(feats, return_outputs_dict=True, noise=z)
    got = nat(feats, return_outputs_dict=True, noise=z)
    assert want['controls']['fdn']['controls']['ir'].shape == (2 * sr,)
    _same(got, want)
    _close(nat(feats, noise=z), want['signal'], 'audio only')
    wet, dry = want['signal'], want['controls']['add']['signal']
    assert (wet - dry).abs().max().item() > 1e-3 * float(dry.abs().max())          # the reverb did something
    if not apply_only:
        fdn.load_parameters({'time_rev_0_sec': 0.25})
        again = py(feats, noise=z)
        assert (again - wet).abs().max().item() > 1e-4 * float(wet.abs().max())
        _close(nat(feats, noise=z), again, 'after load_parameters')
        fast = dp.CapturedGroup(nat, feats)
        _close(fast(feats, noise=z), again, 'graph replay')
    with pytest.raises(ValueError):
        dp.NativeGroup(dp.ProcessorGroup(dp.polyphonic_dag(
            dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True),
            dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr),
            dp.FeedbackDelayNetwork(sampling_rate=sr, name='fdn'), n_synths=P,
            reverb_controls=['input_gain', 'output_gain', 'gain_allpass', 'delays_allpass', 'time_rev_0_sec', 'alpha_tone',
                             'early_ir'], **KEYS)), feats)
