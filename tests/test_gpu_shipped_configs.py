"""Every gin file the reference ships as bench.py builds it for its `shipped_configs` extra: the
group runs at small dims, the batched route takes it (SurrogateAdditive too since round 4), and the batched
route agrees with the node-by-node walk on the same noise."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIGS = {'maestro-v2': (128, 96, 1, 24000, 4800), 'dafx22-24kHz': (128, 96, 2, 24000, 3600),
           'ENSTDkCl-8kHz': (48, 32, 1, 8000, 1600), 'ENSTDkCl-32kHz': (192, 128, 1, 32000, 6400),
           'multi_instruments': (96, 64, 1, 16000, 2400), 'surrogate': (96, 64, 1, 16000, 1600)}


@pytest.mark.parametrize('cfg', sorted(CONFIGS))
def test_shipped_config_group(cfg):
    import bench
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import polyphonic
    H, K, S, sr, L = CONFIGS[cfg]
    B, P, T = 2, 3, 24
    dev = torch.device('cuda', 0)
    feats, _ = bench.make_features(B, P, T, H, K, S, L, dev, seed=5, silent_frac=0.0, midi_lo=40, midi_hi=90)
    if cfg == 'surrogate':
        g = torch.Generator(device=dev)
        g.manual_seed(6)
        for i in range(P):
            feats[f'decays_{i}'] = 0.9990 + 0.0012 * torch.rand(B, T, H, generator=g, device=dev)
            feats[f'decay_time_{i}'] = torch.arange(T, device=dev, dtype=torch.float32).view(1, T, 1).expand(B, T, 1).contiguous()
    N = T * (sr // 250)
    noise = [2.0 * torch.rand(B, N, device=dev) - 1.0 for _ in range(P)]
    pg = bench.build_shipped_group(dp, cfg, P, sr)
    assert polyphonic.recognise(pg.dag) is not None
    out = pg(feats, return_outputs_dict=True, noise=noise)
    assert out['signal'].shape == (B, N) and bool(torch.isfinite(out['signal']).all())
    assert float(out['signal'].abs().max()) > 0.0
    walk = dp.ProcessorGroup(bench.build_shipped_group(dp, cfg, P, sr).dag, fast_path=False)
    if cfg.startswith('ENST'):           # the same network parameters in both groups
        walk.processors[-1].load_parameters(pg.processors[-1].parameters())
    ref = walk(feats, return_outputs_dict=True, noise=noise)['signal']
    scale = max(1.0, float(ref.abs().max()))
    assert float((out['signal'] - ref).abs().max()) < 2e-5 * scale
    if cfg == 'surrogate':               # the dictionary of the batched route holds what the walk's holds for the re-used processors
        walk_out = walk(feats, return_outputs_dict=True, noise=noise)
        for k, v in walk_out['controls']['additive']['controls'].items():
            got = out['controls']['additive']['controls'][k]
            assert got.shape == v.shape and float((got - v).abs().max()) <= 1e-6 * max(1.0, float(v.abs().max())), k
        assert float((out['controls']['additive']['signal'] - walk_out['controls']['additive']['signal']).abs().max()) < 2e-5


@pytest.mark.parametrize('B,P,T,H,sr,vibrato', [
    (3, 5, 130, 96, 16000, 0.0),        # surrogate.gin dims: held notes, spans of 1000 samples start inside frames
    (2, 4, 60, 128, 24000, 0.003),      # moving frequencies (the bank's moving blocks), two oscillators per lane everywhere
    (2, 16, 40, 64, 16000, 0.0),        # sixteen voices, 64 harmonics
])
def test_surrogate_group_against_the_oracle_and_every_route(B, P, T, H, sr, vibrato):
    """configs/surrogate.gin through polyphonic_dag: the compacted bank with the decay term (audio only and the outputs
    dictionary), the every-stem route (per-voice fused kernel) and the node-by-node walk against the numpy oracle."""
    import numpy as np

    import bench
    import ddsp_piano_amd as dp
    from util import O, rms, rms_err
    dev = torch.device('cuda', 0)
    K, L = 64, 1600
    U = sr // 250
    N = T * U
    feats, _ = bench.make_features(B, P, T, H, K, 1, L, dev, seed=9, silent_frac=0.2, midi_lo=30, midi_hi=100, vibrato=vibrato)
    g = torch.Generator(device=dev)
    g.manual_seed(10)
    for i in range(P):
        d = 0.9985 + 0.0017 * torch.rand(B, T, H, generator=g, device=dev)      # some above 1: clipped by get_controls
        feats[f'decays_{i}'] = d
        feats[f'decay_time_{i}'] = (torch.arange(T, device=dev, dtype=torch.float32) % 29).view(1, T, 1).expand(B, T, 1).contiguous()
    noise = [2.0 * torch.rand(B, N, device=dev) - 1.0 for _ in range(P)]
    pg = bench.build_shipped_group(dp, 'surrogate', P, sr)
    out = pg(feats, return_outputs_dict=True, noise=noise)                       # compacted bank, last voice apart
    audio_only = pg(feats, noise=noise)                                          # compacted bank, all voices in one sum
    from ddsp_piano_amd import polyphonic
    stems = polyphonic.run(polyphonic.recognise(pg.dag), feats, noise=noise, need_stems=True)   # per-voice fused kernel
    walk = dp.ProcessorGroup(pg.dag, fast_path=False)(feats, return_outputs_dict=True, noise=noise)
    # oracle
    ofeats = {k: v.cpu().numpy() for k, v in feats.items()}
    additive = O.SurrogateAdditive(name='additive', frame_rate=250, sample_rate=sr, inference=True, scale_fn=O.exp_tanh,
                                   normalize_harm_distribution=False)
    onoise = O.FilteredNoise(name='noise', frame_rate=250, sample_rate=sr, scale_fn=O.exp_tanh)
    dry_ref = np.zeros((B, N), np.float32)
    for i in range(P):
        ctl = additive.get_controls(*[ofeats[f'{k}_{i}'] for k in ('amplitudes', 'decays', 'decay_time', 'harmonic_distribution',
                                                                   'inharm_coef', 'f0_hz')])
        a = additive.get_signal(**ctl)
        z = onoise.get_signal(**onoise.get_controls(ofeats[f'magnitudes_{i}']), noise=noise[i].cpu().numpy())
        dry_ref = (dry_ref + z).astype(np.float32) if i else z.astype(np.float32)
        dry_ref = (dry_ref + a).astype(np.float32)
    tol = 1e-5 * max(1.0, rms(dry_ref))
    for name, got in (('dict', out['controls']['add']['signal']), ('walk', walk['controls']['add']['signal']),
                      ('stems', stems['add']['signal'])):
        assert rms_err(got.cpu().numpy(), dry_ref) < tol, (name, rms_err(got.cpu().numpy(), dry_ref), tol)
    scale = max(1.0, float(walk['signal'].abs().max()))
    assert float((out['signal'] - walk['signal']).abs().max()) < 2e-5 * scale
    assert float((audio_only - walk['signal']).abs().max()) < 2e-5 * scale
    assert float((stems['out']['signal'] - walk['signal']).abs().max()) < 2e-5 * scale
    last = walk['controls']['additive']
    assert float((out['controls']['additive']['signal'] - last['signal']).abs().max()) < 2e-5
    for k, v in last['controls'].items():
        got = out['controls']['additive']['controls'][k]
        assert got.shape == v.shape and float((got - v).abs().max()) <= 1e-6 * max(1.0, float(v.abs().max())), k


def test_surrogate_config_at_bench_size_routes_agree():
    """surrogate.gin at bench.py's size (batch 64 x 3 s, poly 16): the compacted bank with the decay term against the
    per-voice fused kernel, every sample of every row (GPU against GPU; the oracle anchors both at small sizes above)."""
    import bench
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import polyphonic
    dev = torch.device('cuda', 0)
    B, P, T, H, K, sr, L = 64, 16, 750, 96, 64, 16000, 16000
    feats, _ = bench.make_features(B, P, T, H, K, 1, L, dev, seed=43)
    g = torch.Generator(device=dev)
    g.manual_seed(44)
    dec = 0.9990 + 0.0012 * torch.rand(B, P, T, H, generator=g, device=dev)
    dt = torch.arange(T, device=dev, dtype=torch.float32).view(1, 1, T, 1).expand(B, P, T, 1).contiguous()
    for i in range(P):
        feats[f'decays_{i}'], feats[f'decay_time_{i}'] = dec[:, i], dt[:, i]
    N = T * (sr // 250)
    noise = 2.0 * torch.rand(B, P, N, device=dev) - 1.0
    pg = bench.build_shipped_group(dp, 'surrogate', P, sr)
    fast = pg(feats, return_outputs_dict=True, noise=noise)
    stems = polyphonic.run(polyphonic.recognise(pg.dag), feats, noise=noise, need_stems=True)
    ref = stems['out']['signal']
    scale = max(1.0, float(ref.abs().max()))
    assert float((fast['signal'] - ref).abs().max()) < 2e-5 * scale
    assert float((fast['controls']['additive']['signal'] - stems['voices']['additive'][:, P - 1]).abs().max()) < 2e-5
    rms = float(ref.pow(2).mean().sqrt())
    assert rms > 1e-3 and float((fast['signal'] - ref).pow(2).mean().sqrt()) < 1e-6 * max(1.0, rms)


BENCH_SIZE = {'maestro-v2': (128, 96, 1, 24000, 48000), 'dafx22-24kHz': (128, 96, 2, 24000, 36000),
              'ENSTDkCl-8kHz': (48, 32, 1, 8000, 16000), 'ENSTDkCl-32kHz': (192, 128, 1, 32000, 64000),
              'multi_instruments': (96, 64, 1, 16000, 24000), 'surrogate': (96, 64, 1, 16000, 16000)}


@pytest.mark.parametrize('cfg', sorted(BENCH_SIZE))
def test_shipped_config_at_bench_size_against_the_oracle(cfg):
    """Every shipped gin file at the size bench.py times it (batch 64 x 3 s, poly 16, its own flags): the un-reverbed mix
    and the last voice's stems of FOUR segments (the lowest note, the highest, a silent voice, one at random) against the numpy oracle with
    the same flags -- one host thread per voice."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    import bench
    import ddsp_piano_amd as dp
    from util import O, rms, rms_err
    H, K, S, sr, L = BENCH_SIZE[cfg]
    B, P, T = 64, 16, 750
    U = sr // 250
    N = T * U
    dev = torch.device('cuda', 0)
    feats, _ = bench.make_features(B, P, T, H, K, S, L, dev, seed=43)
    if cfg == 'surrogate':
        g = torch.Generator(device=dev)
        g.manual_seed(44)
        dec = 0.9990 + 0.0012 * torch.rand(B, P, T, H, generator=g, device=dev)
        dt = torch.arange(T, device=dev, dtype=torch.float32).view(1, 1, T, 1).expand(B, P, T, 1).contiguous()
        for i in range(P):
            feats[f'decays_{i}'], feats[f'decay_time_{i}'] = dec[:, i], dt[:, i]
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    noise = torch.rand(B, P, N, generator=g, device=dev) * 2.0 - 1.0
    pg = bench.build_shipped_group(dp, cfg, P, sr)
    out = pg(feats, return_outputs_dict=True, noise=noise)
    from test_gpu_full_size import _pick_segments
    segments, info = _pick_segments(feats, P, 4, 5)      # the lowest note, the highest, a silent voice, one at random
    assert info['silent_rows'] > 0
    tanh = cfg in ('ENSTDkCl-8kHz', 'ENSTDkCl-32kHz', 'multi_instruments', 'surrogate')
    scale = {'scale_fn': O.exp_tanh} if tanh else {}
    if cfg == 'surrogate':
        additive = O.SurrogateAdditive(name='additive', frame_rate=250, sample_rate=sr, inference=True,
                                       normalize_harm_distribution=False, **scale)
        akeys = ('amplitudes', 'decays', 'decay_time', 'harmonic_distribution', 'inharm_coef', 'f0_hz')
    else:
        additive = O.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True,
                                     **(dict(normalize_after_nyquist_cut=False) if tanh else {}), **scale)
        akeys = ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')
    onoise = O.FilteredNoise(name='noise', frame_rate=250, sample_rate=sr, **scale)
    rows = torch.as_tensor(segments, device=dev)
    fnp = {k: v[rows].cpu().numpy() for k, v in feats.items() if k != 'reverb_ir'}
    znp = noise[rows].cpu().numpy()

    def voice(task):
        j, i = task
        sl = slice(j, j + 1)
        a = additive.get_signal(**additive.get_controls(*[fnp[f'{k}_{i}'][sl] for k in akeys]))
        z = onoise.get_signal(**onoise.get_controls(fnp[f'magnitudes_{i}'][sl]), noise=znp[sl, i])
        return a, z
    tasks = [(j, i) for j in range(len(segments)) for i in range(P)]
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        sigs = list(ex.map(voice, tasks))
    for j, b in enumerate(segments):
        mix = None
        for a, z in sigs[j * P:(j + 1) * P]:
            mix = (z + a).astype(np.float32) if mix is None else ((mix + z).astype(np.float32) + a).astype(np.float32)
        a_last, z_last = sigs[j * P + P - 1]
        dry = out['controls']['add']['signal'][b:b + 1].cpu().numpy()
        e = rms_err(dry, mix)
        assert e < 1e-5 * max(1.0, rms(mix)), f'{cfg}: segment {b}: dry {e:.3e} vs rms {rms(mix):.3e}'
        assert rms_err(out['controls']['additive']['signal'][b:b + 1].cpu().numpy(), a_last) < 1e-5, (cfg, b)
        assert rms_err(out['controls']['noise']['signal'][b:b + 1].cpu().numpy(), z_last) < 1e-5, (cfg, b)
