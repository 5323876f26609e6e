"""GPU: files longer than 131 072 frames (8.7 min at 24 kHz) -- what synthesize_midi_file.py:41-54,73 renders for most
MAESTRO pieces -- stay on the fused / compacted kernels (round 4).  Past that frame float32(n) * float32(T / N) of the
reference's bilinear resize rounds up to the next whole frame for the last sample(s) of a frame; the kernels get those
samples marked in their weight table (core.walk_weights) and take x[t + 1] exactly."""
import numpy as np
import pytest
import torch

from util import O, rms, rms_err, synth_controls, synth_ir

pytestmark = pytest.mark.gpu
SR, U = 24000, 96
T = 150000                      # 10 minutes
KEYS = dict(additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'], noise_controls=['magnitudes'],
            reverb_controls=['reverb_ir'])


def _voice(rng, H, K=None):
    """One voice of a long file: the pitch steps every ~0.6 s, glides in between, and keeps moving past frame 131 072."""
    c = synth_controls(rng, 1, T, H, S=1, K=K, silent_frac=0.0, midi_lo=40, midi_hi=90)
    steps = 2.0 ** (rng.integers(-3, 4, size=[1, T // 150 + 1, 1]).repeat(150, axis=1)[:, :T] / 12.0)
    c['f0_hz'] = (c['f0_hz'] * steps * (1 + 0.002 * np.sin(np.arange(T) / 9.0))[None, :, None]).astype(np.float32)
    c['amplitudes'] = (c['amplitudes'] * 0 + rng.normal(-1.0, 0.3, [1, T, 1])).astype(np.float32)     # no decay over ten minutes
    return c


def test_long_file_on_the_fused_kernels_equals_the_three_operator_route():
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(150000)
    P, H = 4, 8
    N = T * U
    assert T > core.linear_exact_frames(U) and core.fused_synthesis_supported(T, N)
    assert not core._linear_tables_np(T, N)[3]                      # the plain tables are NOT frame aligned here
    additive = dp.MultiInharmonic(sample_rate=SR, inference=True)
    voices = [_voice(rng, H) for _ in range(P)]
    raw = {k: torch.as_tensor(np.concatenate([v[k] for v in voices], 0), device='cuda') for k in voices[0]}
    ctl = additive._controls(raw['amplitudes'], raw['harmonic_distribution'], raw['inharm_coef'], raw['f0_hz'], want_counts=True)
    amp, hd, sh, f0 = ctl['amplitudes'], ctl['harmonic_distribution'], ctl['harmonic_shifts'], ctl['f0_hz']
    # three operators, one kernel each, from the full (lo, hi, w) tables
    hf = core.get_harmonic_frequencies(f0, H) * (1.0 + sh)
    fe = core.resample(hf, N)
    ae = core.resample(amp * hd, N, method='window')
    want = core.cos_oscillator_bank(fe, ae, SR, True, True)                                   # [P, N]
    del fe, ae
    # (a) the fused kernel, every sample of every voice
    got = core.harmonic_synthesis_fused(f0, amp.reshape(P, T), hd, sh, N, SR, True)
    scale = float(want.abs().max())
    worst = (got - want).abs().max().item() / scale
    assert worst < 2e-6, worst
    first = core.linear_exact_frames(U) * U                         # and in particular around the first marked sample
    sl = slice(first - 1000, first + 1000)
    assert (got[:, sl] - want[:, sl]).abs().max().item() < 2e-6 * scale
    # (b) the compacted bank (what the batched group runs): the voices' sum
    c2 = additive._controls(raw['amplitudes'], raw['harmonic_distribution'], raw['inharm_coef'], raw['f0_hz'], want_counts=True,
                            want_shifts=False)                      # (the bank forms the shifts from inharm_coef itself)
    mix = core.polyphonic_additive(c2['f0_hz'], c2['amplitudes'].reshape(P, T), c2['harmonic_distribution'], None, 1, N, SR,
                                   audible=c2['_audible'], inharm_coef=c2['_inharm_coef'].reshape(P, T))
    ref_mix = want.sum(0, keepdim=True)
    assert (mix - ref_mix).abs().max().item() < 5e-6 * float(ref_mix.abs().max())
    # (c) the oracle, one voice, the whole ten minutes (the phase of sample n depends on every sample before it)
    v = 1
    o = O.MultiInharmonic(sample_rate=SR, inference=True)
    oc = o.get_controls(*[voices[v][k] for k in ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')])
    oref = o.get_signal(**oc)
    g = got[v:v + 1].cpu().numpy()
    assert rms_err(g, oref) < 1e-5 * max(1.0, rms(oref))
    for lo in (first - 1000, first, N - 2000):
        assert rms_err(g[:, lo:lo + 1000], oref[:, lo:lo + 1000]) < 1e-5 * max(1.0, rms(oref))


def test_long_file_through_the_group_and_in_pushes():
    """The batched ProcessorGroup on a ten-minute file, and the same file pushed in three pieces across frame 131 072."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import polyphonic, streaming
    rng = np.random.default_rng(150001)
    P, H, K, L = 4, 16, 96, 24000
    N = T * U
    feats = {}
    for i in range(P):
        for k, v in _voice(rng, H, K).items():
            feats[f'{k}_{i}'] = torch.as_tensor(v, device='cuda')
    feats['reverb_ir'] = torch.as_tensor(synth_ir(rng, 1, L), device='cuda')
    noise = torch.empty((1, P, N), dtype=torch.float32, device='cuda').uniform_(-1, 1, generator=torch.Generator('cuda').manual_seed(3))

    def procs():
        return (dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=SR, inference=True),
                dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=SR), dp.Reverb(name='reverb'))
    pg = dp.ProcessorGroup(dp.polyphonic_dag(*procs(), n_synths=P, **KEYS))
    assert polyphonic.run(pg._plan or polyphonic.recognise(pg.dag), feats, noise=noise, need_stems=False) is not None   # the batched route takes it
    whole = pg(feats, return_outputs_dict=True, noise=noise)
    ref = whole['signal']
    assert ref.shape == (1, N) and torch.isfinite(ref).all()
    walk = dp.ProcessorGroup(pg.dag, fast_path=False)(feats, noise=noise)          # node by node (per-voice fused kernels)
    scale = float(ref.abs().max())
    assert (walk - ref).abs().max().item() < 2e-5 * scale
    syn = streaming.StreamingSynthesizer(*procs(), n_synths=P)
    outs, t0 = [], 0
    for t1 in (50000, 131000, T):                                 # blocks of 125 frames; the second piece crosses frame 131 072
        piece = {k: (v[:, t0:t1] if k != 'reverb_ir' else v) for k, v in feats.items()}
        outs.append(syn.push(piece, noise=noise[:, :, t0 * U:t1 * U], final=(t1 == T)))
        t0 = t1
    got = torch.cat(outs, dim=1)
    assert got.shape == (1, N)
    assert (got - ref).abs().max().item() < 3e-5 * scale, (got - ref).abs().max().item() / scale
