"""GPU: piecewise synthesis with carried state (ddsp_piano_amd/streaming.py) against the one-call render of the same
file -- the whole-file mode of synthesize_midi_file.py:41-73 -- and time ranges rendered from scratch (the unit of a
time shard across GPUs)."""
import numpy as np
import pytest
import torch

from util import synth_controls, synth_ir

pytestmark = pytest.mark.gpu
KEYS = dict(additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'], noise_controls=['magnitudes'],
            reverb_controls=['reverb_ir'])


def _file(rng, B, P, T, H, K, S, L):
    feats = {}
    for i in range(P):
        c = synth_controls(rng, B, T, H, S=S, K=K, silent_frac=0.0, midi_lo=40, midi_hi=96)
        # a file, not a held chord: the pitch steps every ~0.6 s and glides in between
        steps = 2.0 ** (rng.integers(-3, 4, size=[B, T // 150 + 1, 1]).repeat(150, axis=1)[:, :T] / 12.0)
        c['f0_hz'] = (c['f0_hz'] * steps * (1 + 0.002 * np.sin(np.arange(T) / 9.0))[None, :, None]).astype(np.float32)
        for k, v in c.items():
            feats[f'{k}_{i}'] = torch.as_tensor(v, device='cuda')
    feats['reverb_ir'] = torch.as_tensor(synth_ir(rng, B, L), device='cuda')
    return feats


def _processors(dp, sr):
    return (dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True),
            dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr), dp.Reverb(name='reverb'))


@pytest.mark.parametrize('sr,H,K,S', [(24000, 128, 96, 1), (16000, 96, 64, 2), (8000, 48, 64, 1), (8000, 48, 32, 1)])
def test_pushes_equal_the_one_call_render(sr, H, K, S):
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import streaming
    rng = np.random.default_rng(sr)
    B, P, T, L = 1, 4, 1130, 9000
    U = sr // 250
    N = T * U
    feats = _file(rng, B, P, T, H, K, S, L)
    noise = torch.as_tensor(rng.uniform(-1, 1, [B, P, N]).astype(np.float32), device='cuda')
    a, z, r = _processors(dp, sr)
    whole = dp.ProcessorGroup(dp.polyphonic_dag(a, z, r, n_synths=P, **KEYS))(feats, return_outputs_dict=True, noise=noise)
    assert streaming.block_frames(U) == 125
    syn = streaming.StreamingSynthesizer(*_processors(dp, sr), n_synths=P)
    outs, t0 = [], 0
    for t1 in (130, 131, 400, 777, 1000, T):                   # uneven pushes, one of a single frame
        piece = {k: (v[:, t0:t1] if k != 'reverb_ir' else v) for k, v in feats.items()}
        outs.append(syn.push(piece, noise=noise[:, :, t0 * U:t1 * U], final=(t1 == T)))
        t0 = t1
    assert outs[0].shape[1] == 125 * U and outs[1].shape[1] == 0              # whole blocks only, look-ahead kept
    got = torch.cat(outs, dim=1)
    assert got.shape == (B, N)
    ref = whole['signal']
    scale = float(ref.abs().max())
    assert (got - ref).abs().max().item() < 3e-5 * scale, (got - ref).abs().max().item() / scale
    # the carried state is the one-call render's own: the dry mixes agree to summation order
    syn2 = streaming.StreamingSynthesizer(*_processors(dp, sr)[:2], None, n_synths=P)
    dry = torch.cat([syn2.push({k: v[:, :500] for k, v in feats.items() if k != 'reverb_ir'}, noise=noise[:, :, :500 * U]),
                     syn2.push({k: v[:, 500:] for k, v in feats.items() if k != 'reverb_ir'}, noise=noise[:, :, 500 * U:],
                               final=True)], dim=1)
    assert (dry - whole['controls']['add']['signal']).abs().max().item() < 5e-6 * max(1.0, float(dry.abs().max()))


def test_time_ranges_rendered_from_scratch():
    """Two 'ranks' render the two halves of a file independently (phase state from a phase-only pass over the prefix,
    reverb history rendered early): concatenated they are the one-call render."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import parallel, streaming
    rng = np.random.default_rng(4)
    sr, B, P, T, H, K, S, L = 24000, 1, 3, 1000, 128, 96, 1, 20000
    U = sr // 250
    feats = _file(rng, B, P, T, H, K, S, L)
    noise = torch.as_tensor(rng.uniform(-1, 1, [B, P, T * U]).astype(np.float32), device='cuda')
    a, z, r = _processors(dp, sr)
    whole = dp.ProcessorGroup(dp.polyphonic_dag(a, z, r, n_synths=P, **KEYS))(feats, noise=noise)

    def make():
        return streaming.StreamingSynthesizer(*_processors(dp, sr), n_synths=P)
    ranges = [parallel.time_shard_range(T, 2, rk, 125) for rk in range(2)]
    assert ranges == [(0, 500), (500, 1000)]
    parts = [streaming.render_range(make, feats, lo, hi, noise=noise) for lo, hi in ranges]
    got = torch.cat(parts, dim=1)
    assert got.shape == whole.shape
    assert (got - whole).abs().max().item() < 3e-5 * float(whole.abs().max())
    with pytest.raises(ValueError):
        streaming.render_range(make, feats, 60, 500, noise=noise)
    # the library's own noise stream is addressed by absolute position: any split renders the same file
    one = streaming.render_range(make, feats, 0, T)
    two = torch.cat([streaming.render_range(make, feats, 0, 375), streaming.render_range(make, feats, 375, T)], dim=1)
    assert (one - two).abs().max().item() < 3e-5 * float(one.abs().max())


def test_noise_context_at_8khz_reaches_two_frames():
    """ENSTDkCl-8kHz with 64 noise bands: hop 32, FIR of 126 taps advanced by 61 -- it reaches 64 samples back and 61
    ahead, two frames each way.  Ranges rendered from scratch equal the one-call render, with explicit noise
    and with the library's position-keyed stream."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import streaming
    assert streaming.noise_reach(64, 257, 32) == (2, 2) and streaming.noise_reach(96, 257, 96) == (1, 1)
    assert streaming.noise_reach(64, 257, 64) == (1, 1) and streaming.noise_reach(32, 257, 32) == (1, 1)
    rng = np.random.default_rng(8)
    sr, B, P, T, H, K, S, L = 8000, 2, 3, 750, 48, 64, 1, 3000
    U = sr // 250
    feats = _file(rng, B, P, T, H, K, S, L)
    noise = torch.as_tensor(rng.uniform(-1, 1, [B, P, T * U]).astype(np.float32), device='cuda')
    a, z, r = _processors(dp, sr)
    whole = dp.ProcessorGroup(dp.polyphonic_dag(a, z, r, n_synths=P, **KEYS))(feats, noise=noise)

    def make():
        return streaming.StreamingSynthesizer(*_processors(dp, sr), n_synths=P)
    parts = [streaming.render_range(make, feats, lo, hi, noise=noise) for lo, hi in ((0, 250), (250, 625), (625, T))]
    got = torch.cat(parts, dim=1)
    assert got.shape == whole.shape
    assert (got - whole).abs().max().item() < 3e-5 * float(whole.abs().max())
    one = streaming.render_range(make, feats, 0, T)
    two = torch.cat([streaming.render_range(make, feats, 0, 375), streaming.render_range(make, feats, 375, T)], dim=1)
    assert (one - two).abs().max().item() < 3e-5 * float(one.abs().max())


def test_state_of_oscillators_that_are_silent_for_a_whole_piece():
    """A partial that is above Nyquist for a whole piece still advances its phase in the reference (angular_cumsum runs
    on every frequency envelope, inharm_synth.py:65-75).  Many rows (the memoised phase walk, which skips silent
    64-partial groups inside ONE call): block 1 glides at a high pitch, block 2 holds it, block 3 drops seven octaves
    lower partials 64..127 come in -- with the phase they accumulated during blocks 1 and 2."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import streaming
    rng = np.random.default_rng(77)
    sr, B, P, T, H, K, S = 24000, 16, 16, 375, 128, 96, 1
    U = sr // 250
    feats = {}
    t = np.arange(T)
    pitch = np.where(t < 125, 1000.0 * (1.0 + 0.1 * t / 125.0), np.where(t < 250, 1100.0, 55.0))
    for i in range(P):
        c = synth_controls(rng, B, T, H, S=S, K=K, silent_frac=0.0)
        c['f0_hz'] = (pitch[None, :, None] * (1.0 + 0.01 * rng.random([B, 1, 1]))).astype(np.float32) * np.ones([1, 1, S], np.float32)
        c['inharm_coef'] = np.full([B, T, 1], 1e-4, np.float32)
        c['amplitudes'] = np.zeros([B, T, 1], np.float32)
        for k, v in c.items():
            feats[f'{k}_{i}'] = torch.as_tensor(v, device='cuda')
    noise = torch.zeros([B, P, T * U], device='cuda')
    a, z, _ = _processors(dp, sr)
    whole = dp.ProcessorGroup(dp.polyphonic_dag(a, z, None, n_synths=P, **{**KEYS, 'reverb_controls': []}))(
        feats, noise=noise)
    syn = streaming.StreamingSynthesizer(*_processors(dp, sr)[:2], None, n_synths=P)
    outs, t0 = [], 0
    for t1 in (126, 251, T):
        outs.append(syn.push({k: v[:, t0:t1] for k, v in feats.items()}, noise=noise[:, :, t0 * U:t1 * U], final=(t1 == T)))
        t0 = t1
    got = torch.cat(outs, dim=1)
    assert got.shape == whole.shape
    err = (got - whole)[:, 250 * U:].abs().max().item()
    assert err < 1e-6 * float(whole[:, 250 * U:].abs().max()), err
