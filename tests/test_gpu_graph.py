"""GPU: a ProcessorGroup call captured as a HIP graph (ddsp_piano_amd/graph.py) replays the eager call bit for bit on
new inputs of the same shape, and draws fresh noise on every replay."""
import numpy as np
import pytest
import torch

from util import synth_controls, synth_ir

pytestmark = pytest.mark.gpu
KEYS = dict(additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'], noise_controls=['magnitudes'],
            reverb_controls=['reverb_ir'])


def _features(seed, B, P, T, H, K, S, L):
    """Per-voice keys as views of one [B, P, T, C] buffer per control (how a batched control network hands them over)."""
    rng = np.random.default_rng(seed)
    voices = [synth_controls(rng, B, T, H, S=S, K=K, silent_frac=0.2) for _ in range(P)]
    feats = {}
    for k in voices[0]:
        whole = torch.as_tensor(np.stack([v[k] for v in voices], axis=1), device='cuda')      # [B, P, T, C]
        for i in range(P):
            feats[f'{k}_{i}'] = whole[:, i]
    feats['reverb_ir'] = torch.as_tensor(synth_ir(rng, B, L), device='cuda')
    return feats


@pytest.mark.parametrize('dict_form', [False, True])
def test_replay_equals_the_eager_call(dict_form):
    import ddsp_piano_amd as dp
    sr, B, P, T, H, K, S, L = 24000, 2, 3, 125, 128, 96, 1, 6000
    N = T * (sr // 250)
    group = dp.ProcessorGroup(dp.polyphonic_dag(
        dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True),
        dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr), dp.Reverb(name='reverb'),
        n_synths=P, **KEYS))
    fast = dp.CapturedGroup(group, _features(1, B, P, T, H, K, S, L), return_outputs_dict=dict_form)
    rng = np.random.default_rng(9)
    for seed in (2, 3):
        feats = _features(seed, B, P, T, H, K, S, L)
        noise = torch.as_tensor(rng.uniform(-1, 1, [B, P, N]).astype(np.float32), device='cuda')
        want = group(feats, return_outputs_dict=dict_form, noise=noise)
        got = fast(feats, noise=noise)
        if dict_form:
            assert torch.equal(got['signal'], want['signal'])
            assert torch.equal(got['controls']['additive']['signal'], want['controls']['additive']['signal'])
            assert torch.equal(got['controls']['add']['signal'], want['controls']['add']['signal'])
        else:
            assert torch.equal(got, want)
    # the library's own noise: a new draw per replay (a captured generator call would repeat itself)
    a = fast(feats)
    a = (a['signal'] if dict_form else a).clone()
    b = fast(feats)
    b = b['signal'] if dict_form else b
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert (a - b).abs().max().item() > 1e-4
    with pytest.raises(ValueError):
        fast({k: v[:, :100] if k != 'reverb_ir' else v for k, v in feats.items()})


def test_native_group_captured_with_inputs_written_in_place():
    """The one-call driver's kernels as a graph, controls written straight into the captured buffers (`fast.inputs`,
    `fast()`): no per-replay copies; equal to the eager NativeGroup call on the same inputs and noise."""
    import ddsp_piano_amd as dp
    sr, B, P, T, H, K, S, L = 24000, 1, 4, 125, 128, 96, 1, 6000
    N = T * (sr // 250)
    group = dp.ProcessorGroup(dp.polyphonic_dag(
        dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True),
        dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr), dp.Reverb(name='reverb'),
        n_synths=P, **KEYS))
    f0 = _features(1, B, P, T, H, K, S, L)
    native = dp.NativeGroup(group, f0)
    fast = dp.CapturedGroup(native, f0, return_outputs_dict=True)
    assert set(fast.inputs) == set(f0)
    rng = np.random.default_rng(4)
    for seed in (5, 6):
        feats = _features(seed, B, P, T, H, K, S, L)
        noise = torch.as_tensor(rng.uniform(-1, 1, [B, P, N]).astype(np.float32), device='cuda')
        want = native(feats, return_outputs_dict=True, noise=noise)
        for k, v in feats.items():
            fast.inputs[k].copy_(v)                   # the caller's producer writes here
        got = fast(noise=noise)
        assert torch.equal(got['signal'], want['signal'])
        assert torch.equal(got['controls']['add']['signal'], want['controls']['add']['signal'])
        same = fast(fast.inputs, noise=noise)          # passing the captured buffers themselves: nothing is copied either
        assert torch.equal(same['signal'], want['signal'])
    a = fast()['signal'].clone()                      # the library's own noise stream: a fresh draw per replay
    b = fast()['signal'].clone()
    assert not torch.equal(a, b) and torch.isfinite(a).all()


def test_replay_with_the_large_lds_noise_instances():
    """The 32 kHz / 48 kHz FilteredNoise instances raise their dynamic-LDS limit at every launch (per device):
    that has to be legal inside a stream capture too.  48 kHz dims (hop 192, K = 96) and 32 kHz (hop 128, K = 128)."""
    import ddsp_piano_amd as dp
    for sr, H, K in ((48000, 64, 96), (32000, 64, 128)):
        B, P, T, S, L = 2, 2, 125, 1, 4000
        N = T * (sr // 250)
        group = dp.ProcessorGroup(dp.polyphonic_dag(
            dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True),
            dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr), dp.Reverb(name='reverb'),
            n_synths=P, **KEYS))
        fast = dp.CapturedGroup(group, _features(1, B, P, T, H, K, S, L))
        feats = _features(2, B, P, T, H, K, S, L)
        noise = torch.as_tensor(np.random.default_rng(sr).uniform(-1, 1, [B, P, N]).astype(np.float32), device='cuda')
        want = group(feats, noise=noise)
        got = fast(feats, noise=noise)
        assert torch.equal(got, want), sr


def test_replay_when_the_frequencies_start_and_stop_moving():
    """Which kernels a call launches may depend on shapes and options only, never on the data: a graph captured on held
    notes has to render a vibrato on replay (the compacted scan of moving chunks is decided by a word the count kernel
    writes, inside the captured launches), and held notes again after that (the word of the previous replay is still in the
    workspace).  A batch large enough for the memoised pre-pass + bank_scan_kernel (256 rows)."""
    import ddsp_piano_amd as dp
    sr, B, P, T, H, K, S, L = 24000, 64, 4, 120, 128, 96, 1, 3000
    N = T * (sr // 250)
    group = dp.ProcessorGroup(dp.polyphonic_dag(
        dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True),
        dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr), dp.Reverb(name='reverb'),
        n_synths=P, **KEYS))
    held = _features(11, B, P, T, H, K, S, L)
    fast = dp.CapturedGroup(group, held)
    tt = torch.arange(T, device='cuda', dtype=torch.float32)[None, :, None]
    noise = torch.as_tensor(np.random.default_rng(12).uniform(-1, 1, [B, P, N]).astype(np.float32), device='cuda')
    outs = {}
    for step, moving in enumerate((True, False, True, False)):
        feats = dict(_features(20 + step, B, P, T, H, K, S, L))
        if moving:
            for i in range(P):
                feats[f'f0_hz_{i}'] = feats[f'f0_hz_{i}'] * (1 + 0.004 * torch.sin(0.13 * tt + i))
        want = group(feats, noise=noise)
        got = fast(feats, noise=noise)
        assert torch.equal(got, want), (step, moving)
        outs[step] = want.clone()
    assert not torch.equal(outs[0], outs[1])
