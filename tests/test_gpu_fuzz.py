"""GPU: randomised route cross-checks on 'musical' controls -- notes that start, stop, change pitch and glide at random
frames, silent stretches, inharmonicity that follows the note -- so that the data-dependent shortcuts of the oscillator
bank (silent groups, held-note memo, constant-frequency blocks, Nyquist crossings, audible-partial compaction) switch
on and off inside one call.  The compacted polyphonic bank must equal the sum of the per-voice stems of the fused
kernel (same phases bit for bit: only the summation order differs), with and without the per-frame counts, for several
span decompositions; small cases are also held against the oracle."""
import numpy as np
import pytest
import torch

from util import O, rms_err

pytestmark = pytest.mark.gpu


def musical_controls(rng, R, T, H, S, sr):
    """Raw controls [R, T, .]: every row is a voice playing a random sequence of notes / rests."""
    f0 = np.zeros([R, T, S], np.float32)
    inh = np.zeros([R, T, 1], np.float32)
    amp = np.zeros([R, T, 1], np.float32)
    detune = 2.0 ** (0.4 * np.arange(S) / 1200.0)
    for r in range(R):
        t = 0
        while t < T:
            dur = int(rng.integers(1, max(2, T // 2)))
            kind = rng.random()
            sl = slice(t, min(t + dur, T))
            n = sl.stop - sl.start
            if kind < 0.25:                                  # rest: f0 = 0, the voice is gated off
                pass
            else:
                midi = rng.uniform(21, 108)
                hz = 440.0 * 2.0 ** ((midi - 69.0) / 12.0)
                if kind < 0.45:                              # glide / vibrato: every frame moves
                    bend = 2.0 ** (rng.uniform(-2, 2) * np.linspace(0, 1, n) / 12.0) * (1 + 0.003 * np.sin(np.arange(n) / 3.0))
                else:                                        # held note
                    bend = np.ones(n)
                f0[r, sl, :] = (hz * bend)[:, None] * detune[None, :]
                inh[r, sl, 0] = np.exp(-0.105 * midi - 6.87) + np.exp(0.094 * midi - 13.70)
                amp[r, sl, 0] = rng.normal(-1.0, 1.0) - 3.0 * np.linspace(0, 1, n)
            t += dur
    hd = (rng.normal(0.0, 1.0, [R, T, H]) - 0.04 * np.arange(1, H + 1)[None, None, :]).astype(np.float32)
    return dict(amplitudes=amp, harmonic_distribution=hd, inharm_coef=inh, f0_hz=f0)


CASES = [  # (seed, B, P, T, H, S, U, flags)
    (1, 1, 4, 60, 128, 1, 96, {}), (2, 2, 3, 130, 96, 2, 64, {}), (3, 3, 5, 47, 64, 1, 128, {}), (4, 1, 16, 260, 128, 1, 96, {}),
    (5, 16, 16, 64, 128, 1, 96, {}),            # 256 rows: the memoised pre-pass (and its silent-group exit)
    (6, 20, 16, 33, 96, 2, 64, {}), (7, 2, 2, 700, 48, 1, 32, {}), (8, 1, 1, 300, 192, 1, 128, {}), (9, 5, 7, 90, 16, 1, 192, {}),
    # the ENSTDkCl configurations' flags (ENSTDkCl-8kHz.gin / -32kHz.gin): exp_tanh, no renormalisation after the cut
    (31, 2, 16, 120, 48, 1, 32, dict(scale='exp_tanh', normalize_after_nyquist_cut=False)),
    (32, 1, 8, 90, 192, 1, 128, dict(scale='exp_tanh', normalize_after_nyquist_cut=False)),
    # partials are NOT cut at the frame rate: only the per-sample mask of the oscillator bank silences them
    (33, 2, 5, 80, 128, 1, 96, dict(normalize_below_nyquist=False)),
    (34, 17, 16, 50, 128, 1, 96, dict(normalize_below_nyquist=False, normalize_after_nyquist_cut=False)),
]


@pytest.mark.parametrize('seed,B,P,T,H,S,U,flags', CASES)
def test_compacted_bank_equals_the_stems(seed, B, P, T, H, S, U, flags):
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(seed)
    sr = 250 * U
    R, N = B * P, T * U
    raw = musical_controls(rng, R, T, H, S, sr)
    flags = dict(flags)
    scale_name = flags.pop('scale', 'exp_sigmoid')
    add = dp.MultiInharmonic(sample_rate=sr, inference=True, scale_fn=getattr(dp, scale_name), **flags)
    dev = [torch.as_tensor(raw[k], device='cuda') for k in ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')]
    ctl = add._controls(*dev, want_counts=True)
    amp = ctl['amplitudes'].reshape(R, T).contiguous()
    stems = core.harmonic_synthesis_fused(ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'], N, sr,
                                          True).reshape(B, P, N)
    want = stems.sum(1)
    scale = max(1.0, float(want.abs().max()))
    inh = dev[2].reshape(R, T).contiguous()            # raw inharm_coef: the kernels form the shifts themselves
    outs = {}
    for name, kw in (('counts', dict(audible=ctl['_audible'])), ('no counts', {}),
                     ('spans=1', dict(audible=ctl['_audible'], spans=1)), ('spans=3', dict(audible=ctl['_audible'], spans=3)),
                     ('from inharm_coef', dict(audible=ctl['_audible'], shifts=None, inharm_coef=inh))):
        shifts = kw.pop('shifts', ctl['harmonic_shifts'])
        outs[name] = core.polyphonic_additive(ctl['f0_hz'], amp, ctl['harmonic_distribution'], shifts, B, N, sr, **kw)
        assert (outs[name] - want).abs().max().item() < 6e-6 * scale, name
    # (another span decomposition packs other oscillators into a slot: same phases, another summation order)
    # shifts formed in the kernel from inharm_coef are the get_controls kernel's, bit for bit
    assert torch.equal(outs['counts'], outs['from inharm_coef'])
    # split_last: the last voice on its own + the others
    rest, last = core.polyphonic_additive(ctl['f0_hz'], amp, ctl['harmonic_distribution'], None, B, N, sr,
                                          audible=ctl['_audible'], inharm_coef=inh, split_last=True)
    assert (last - stems[:, P - 1]).abs().max().item() < 6e-6 * scale
    assert ((rest + last) - want).abs().max().item() < 6e-6 * scale
    # every voice's stem from the same packing with the harmonic sum stopped at voice boundaries (ddspp_polyphonic_stems)
    flat = stems.reshape(R, N)
    sscale = max(1.0, float(flat.abs().max()))
    for name, kw in (('counts', dict(audible=ctl['_audible'])), ('no counts', {}), ('spans=1', dict(audible=ctl['_audible'], spans=1)),
                     ('spans=3, shifts given', dict(audible=ctl['_audible'], spans=3, shifts=ctl['harmonic_shifts']))):
        shifts = kw.pop('shifts', None)
        got = core.polyphonic_stems(ctl['f0_hz'], amp, ctl['harmonic_distribution'], shifts, B, N, sr,
                                    inharm_coef=None if shifts is not None else inh, **kw)
        assert (got - flat).abs().max().item() < 4e-6 * sscale, ('stems', name)
    if R * N * H <= 16 * 36000 * 128:                      # small enough for the numpy oracle: a few seconds
        ref = O.MultiInharmonic(sample_rate=sr, inference=True, scale_fn=getattr(O, scale_name), **flags)(**raw)
        assert rms_err(got.cpu().numpy(), ref) < 2e-6
        ref = ref.reshape(B, P, N).sum(1)
        assert rms_err(outs['counts'].cpu().numpy(), ref) < 2e-6


@pytest.mark.parametrize('seed,B,P,T,H,K,S,U,vm', [(11, 2, 4, 140, 128, 96, 1, 96, False), (12, 3, 3, 75, 96, 64, 2, 64, True),
                                                   (13, 1, 16, 250, 128, 96, 1, 96, False), (14, 17, 16, 40, 128, 96, 1, 96, True)])
def test_batched_group_equals_the_node_by_node_walk(seed, B, P, T, H, K, S, U, vm, monkeypatch, surrogate=False):
    """(surrogate=True, tools/fuzz_soak.py and the test below: SurrogateAdditive voices -- configs/surrogate.gin -- with random
    decay factors, the compacted bank's decay variant against the walk.)
    The batched route (compacted bank, fused noise with voice sums, split last voice, early IR transform) against the
    DAG walked node by node through the per-processor entry points, both call forms, on musical controls; vm: the
    per-voice keys are views of one voice-major [P, B, T, C] buffer (the Parallelizer's un-merge) instead of [B, P, T, C]."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(seed)
    sr = 250 * U
    N, L = T * U, 3000
    raw = musical_controls(rng, B * P, T, H, S, sr)
    raw['magnitudes'] = rng.normal(0.0, 1.5, [B * P, T, K]).astype(np.float32)
    feats = {}
    for k, v in raw.items():
        whole = torch.as_tensor(v.reshape(*((P, B) if vm else (B, P)), T, v.shape[-1]), device='cuda')
        for i in range(P):
            feats[f'{k}_{i}'] = whole[i] if vm else whole[:, i]
    ir = rng.normal(0.0, 1.0, [B, L]) * np.exp(-6.9 * np.arange(L) / L)[None, :] * 0.05
    feats['reverb_ir'] = torch.as_tensor(ir.astype(np.float32), device='cuda')
    noise = torch.as_tensor(rng.uniform(-1, 1, [B, P, N]).astype(np.float32), device='cuda')
    keys = dict(additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'],
                noise_controls=['magnitudes'], reverb_controls=['reverb_ir'])
    if surrogate:
        assert S == 1
        keys['additive_controls'] = ['amplitudes', 'decays', 'decay_time', 'harmonic_distribution', 'inharm_coef', 'f0_hz']
        dec = rng.uniform(0.998, 1.0003, [*((P, B) if vm else (B, P)), T, H]).astype(np.float32)
        dec[..., ::5] *= -1.0
        dtm = np.broadcast_to((np.arange(T, dtype=np.float32) % int(rng.integers(3, 60)))[:, None], dec.shape[:2] + (T, 1)).copy()
        dec_t, dtm_t = torch.as_tensor(dec, device='cuda'), torch.as_tensor(dtm, device='cuda')
        for i in range(P):
            feats[f'decays_{i}'] = dec_t[i] if vm else dec_t[:, i]
            feats[f'decay_time_{i}'] = dtm_t[i] if vm else dtm_t[:, i]

    def group(fast):
        if surrogate:
            return dp.ProcessorGroup(dp.polyphonic_dag(
                dp.SurrogateAdditive(name='additive', frame_rate=250, sample_rate=sr, inference=True,
                                     normalize_harm_distribution=bool(seed % 3)),
                dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr), dp.Reverb(name='reverb'),
                n_synths=P, **keys), fast_path=fast)
        return dp.ProcessorGroup(dp.polyphonic_dag(
            dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True),
            dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr), dp.Reverb(name='reverb'),
            n_synths=P, **keys), fast_path=fast)
    slow = group(False)(feats, return_outputs_dict=True, noise=noise)
    # small batches keep per-voice noise rows (workgroups enough without voice sums); seeds divisible by 2 force the sums
    # (of 8 / 4 / 2 voices, the last voice split off inside the kernel) so that both forms are walked at test sizes
    from util import set_option
    if seed % 2 == 0:
        set_option(monkeypatch, 'DDSPP_VOICE_SUMS', 8)
    fast = group(True)(feats, return_outputs_dict=True, noise=noise)
    audio = group(True)(feats, noise=noise)
    if seed % 2 == 0:
        set_option(monkeypatch, 'DDSPP_VOICE_SUMS')
    scale = max(1.0, float(slow['signal'].abs().max()))
    assert (fast['signal'] - slow['signal']).abs().max().item() < 2e-5 * scale
    assert (audio - slow['signal']).abs().max().item() < 2e-5 * scale
    for node in ('additive', 'noise', 'add'):                 # the last voice's stems and the dry mix
        a, b = fast['controls'][node]['signal'], slow['controls'][node]['signal']
        assert (a - b).abs().max().item() < 1e-5 * max(1.0, float(b.abs().max())), node
    for k in ('amplitudes', 'harmonic_distribution', 'harmonic_shifts', 'f0_hz') + (('decays', 'decay_time') if surrogate else ()):
        a, b = fast['controls']['additive']['controls'][k], slow['controls']['additive']['controls'][k]
        assert a.shape == b.shape and (a - b).abs().max().item() <= 1e-6 * max(1.0, float(b.abs().max())), k


@pytest.mark.parametrize('seed,B,P,T,H,K,U,vm', [(31, 2, 4, 140, 96, 64, 64, False), (32, 3, 16, 75, 128, 96, 96, True),
                                                 (33, 1, 5, 300, 48, 32, 32, False), (34, 5, 7, 60, 192, 96, 192, True)])
def test_batched_surrogate_group_equals_the_node_by_node_walk(seed, B, P, T, H, K, U, vm, monkeypatch):
    test_batched_group_equals_the_node_by_node_walk(seed, B, P, T, H, K, 1, U, vm, monkeypatch, surrogate=True)


@pytest.mark.parametrize('seed,B,P,T,H,K,S,U', [(21, 1, 6, 500, 128, 96, 1, 96), (22, 16, 16, 375, 128, 96, 1, 96),
                                               (23, 2, 3, 500, 96, 64, 2, 64)])
def test_streamed_pieces_equal_the_one_call_render(seed, B, P, T, H, K, S, U):
    """Pieces of a file pushed at random cuts (carried oscillator state, noise context, reverb overlap) against the
    whole file in one call, musical controls: notes change inside and across pieces, partials come and go."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import streaming
    rng = np.random.default_rng(seed)
    sr = 250 * U
    N, L = T * U, 5000
    raw = musical_controls(rng, B * P, T, H, S, sr)
    raw['magnitudes'] = rng.normal(0.0, 1.5, [B * P, T, K]).astype(np.float32)
    feats = {}
    for k, v in raw.items():
        whole = torch.as_tensor(v.reshape(B, P, T, v.shape[-1]), device='cuda')
        for i in range(P):
            feats[f'{k}_{i}'] = whole[:, i]
    ir = rng.normal(0.0, 1.0, [B, L]) * np.exp(-6.9 * np.arange(L) / L)[None, :] * 0.05
    feats['reverb_ir'] = torch.as_tensor(ir.astype(np.float32), device='cuda')
    noise = torch.as_tensor(rng.uniform(-1, 1, [B, P, N]).astype(np.float32), device='cuda')
    keys = dict(additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'],
                noise_controls=['magnitudes'], reverb_controls=['reverb_ir'])

    def procs():
        return (dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True),
                dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr), dp.Reverb(name='reverb'))
    whole = dp.ProcessorGroup(dp.polyphonic_dag(*procs(), n_synths=P, **keys))(feats, return_outputs_dict=True, noise=noise)
    syn = streaming.StreamingSynthesizer(*procs(), n_synths=P)
    cuts = sorted(set(int(c) for c in rng.integers(1, T, size=4))) + [T]
    outs, t0 = [], 0
    for t1 in cuts:
        piece = {k: (v[:, t0:t1] if k != 'reverb_ir' else v) for k, v in feats.items()}
        outs.append(syn.push(piece, noise=noise[:, :, t0 * U:t1 * U], final=(t1 == T)))
        t0 = t1
    got = torch.cat(outs, dim=1)
    assert got.shape == whole['signal'].shape
    scale = max(1.0, float(whole['signal'].abs().max()))
    assert (got - whole['signal']).abs().max().item() < 3e-5 * scale
