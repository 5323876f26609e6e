"""GPU parity: the whole polyphonic ProcessorGroup (batched route and node-by-node route) vs oracle."""
import numpy as np
import pytest
import torch

from util import O, rms, rms_err, set_option, synth_controls, synth_ir

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _features(rng, B, P, T, H, K, S, L):
    feats = {}
    for i in range(P):
        c = synth_controls(rng, B, T, H, S=S, K=K)
        for k, v in c.items():
            feats[f'{k}_{i}'] = v
    feats['reverb_ir'] = synth_ir(rng, B, L)
    return feats


def _build(mod, P, sr, inference=True, scale=None, with_reverb=True):
    kw = {}
    if scale is not None:
        kw['scale_fn'] = scale
    additive = mod.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=inference, **kw)
    if mod is O:
        noise = mod.FilteredNoise(name='noise', frame_rate=250, sample_rate=sr, **kw)
    else:
        noise = mod.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr, **kw)
    reverb = mod.Reverb(name='reverb') if with_reverb else None
    dag = mod.polyphonic_dag(additive, noise, reverb,
                             additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'],
                             noise_controls=['magnitudes'], reverb_controls=['reverb_ir'] if with_reverb else [],
                             n_synths=P)
    return dag, noise


@pytest.mark.parametrize('B,P,T,H,K,S,sr,L', [
    (2, 3, 50, 128, 96, 1, 24000, 4000),     # maestro-v2 dims, down-sized
    (1, 2, 50, 96, 64, 2, 16000, 3000),      # dafx22 dims (two sub-strings), SURVEY 8c down-sized C2
    (2, 16, 20, 96, 64, 1, 24000, 2000),     # full polyphony
])
def test_processor_group_matches_oracle(B, P, T, H, K, S, sr, L):
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(1234)
    feats = _features(rng, B, P, T, H, K, S, L)
    N = T * (sr // 250)
    noises = [rng.uniform(-1, 1, [B, N]).astype(np.float32) for _ in range(P)]

    odag, _ = _build(O, P, sr)
    oref = O.ProcessorGroup(odag)(feats, return_outputs_dict=True,
                                  extra_kwargs={'noise': [{'noise': z} for z in noises]})
    gfeats = {k: torch.as_tensor(v, device='cuda') for k, v in feats.items()}

    for fast in (True, False):
        gdag, gnoise = _build(dp, P, sr)
        pg = dp.ProcessorGroup(gdag, fast_path=fast)
        out = pg(gfeats, return_outputs_dict=True, noise=[torch.as_tensor(z, device='cuda') for z in noises])
        sig = out['signal'].cpu().numpy()
        ref = oref['signal']
        assert sig.shape == ref.shape == (B, N)
        err = rms_err(sig, ref)
        assert err < TOL * max(1.0, rms(ref)), f'fast={fast}: {err:.3e} vs rms {rms(ref):.3e}'
        ctl = out['controls']
        dry = ctl['add']['signal'].cpu().numpy()
        assert rms_err(dry, oref['controls']['add']['signal']) < TOL
        # the reused processors leave the LAST voice behind (polyphonic_dag.py re-uses three objects)
        assert rms_err(ctl['additive']['signal'].cpu().numpy(), oref['controls']['additive']['signal']) < TOL
        assert rms_err(ctl['noise']['signal'].cpu().numpy(), oref['controls']['noise']['signal']) < TOL
        assert 'reverb_ir' in ctl and 'amplitudes_0' in ctl and ctl['out'] is ctl['reverb']
        assert [p.name for p in pg.processors] == ['additive', 'noise', 'add', 'reverb']


@pytest.mark.parametrize('voice_major', [False, True])
@pytest.mark.parametrize('stems', [False, True])
def test_fast_path_zero_copy_views_and_no_reverb(voice_major, stems):
    """Per-voice keys that are slices of one buffer are taken without a copy, in both orders: segment major
    [B, P, T, C] and the reference Parallelizer's voice major [P, B, T, C] (sub_modules.py:573-592)."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import polyphonic
    rng = np.random.default_rng(5)
    B, P, T, H, K, sr = 3, 4, 20, 64, 32, 16000
    lead = (P, B) if voice_major else (B, P)
    base = {k: torch.as_tensor(rng.normal(0, 1, [*lead, T, c]).astype(np.float32), device='cuda')
            for k, c in (('amplitudes', 1), ('harmonic_distribution', H), ('magnitudes', K))}
    base['inharm_coef'] = torch.full((*lead, T, 1), 1e-4, device='cuda')
    base['f0_hz'] = torch.as_tensor(rng.uniform(50, 2000, [*lead, 1, 1]).astype(np.float32), device='cuda').expand(*lead, T, 1).contiguous()
    feats = {f'{k}_{i}': (v[i] if voice_major else v[:, i]) for k, v in base.items() for i in range(P)}
    stacked, vm = polyphonic._stack_voices([feats[f'harmonic_distribution_{i}'] for i in range(P)])
    assert stacked.data_ptr() == base['harmonic_distribution'].data_ptr() and vm == voice_major   # no copy made
    assert stacked.shape == (B * P, T, H)
    dag, _ = _build(dp, P, sr, with_reverb=False)
    a = dp.ProcessorGroup(dag, fast_path=True)
    dagb, _ = _build(dp, P, sr, with_reverb=False)
    b = dp.ProcessorGroup(dagb, fast_path=False)
    noise = torch.as_tensor(rng.uniform(-1, 1, [P, B, T * 64]).astype(np.float32), device='cuda')
    nz = [noise[i] for i in range(P)]
    if stems:
        oa, ob = a(feats, return_outputs_dict=True, need_stems=True, noise=nz), b(feats, return_outputs_dict=True, noise=nz)
        ya, yb = oa['signal'], ob['signal']
        for name in ('additive', 'noise'):             # the last voice's stems, as the node-by-node walk leaves them
            assert (oa['controls'][name]['signal'] - ob['controls'][name]['signal']).abs().max().item() < 2e-6
        voices = oa['controls']['voices']['additive']
        assert voices.shape == (B, P, T * 64)
        assert torch.equal(voices[:, P - 1], oa['controls']['additive']['signal'])
    else:
        ya, yb = a(feats, noise=nz), b(feats, noise=noise.transpose(0, 1))      # list of P [B, N] == tensor [B, P, N]
    assert ya.shape == (B, T * 64)
    assert (ya - yb).abs().max().item() < 2e-6


def test_stack_voices_copies_when_layouts_disagree():
    from ddsp_piano_amd import polyphonic
    B, P, T, C = 2, 3, 5, 4
    x = torch.arange(B * P * T * C, dtype=torch.float32, device='cuda').reshape(B, P, T, C)
    sep = [x[:, i].clone() for i in range(P)]
    for vm in (True, False):
        got, flag = polyphonic._stack_voices(sep, vm)
        want = x.transpose(0, 1).reshape(P * B, T, C) if vm else x.reshape(B * P, T, C)
        assert flag == vm and torch.equal(got, want)
    # a segment-major view asked for in voice-major order has to be re-laid out
    got, flag = polyphonic._stack_voices([x[:, i] for i in range(P)], True)
    assert flag is True and torch.equal(got, x.transpose(0, 1).reshape(P * B, T, C))


def test_standalone_processors_like_synthesize_from_csv():
    """synthesize_from_csv.py:99-120 calls processors[:2] outside the group."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(9)
    dag, _ = _build(dp, 2, 24000)
    pg = dp.ProcessorGroup(dag)
    additive, noise = pg.processors[:2]
    c = {k: torch.as_tensor(v, device='cuda') for k, v in synth_controls(rng, 1, 30, 128, K=96).items()}
    sig = additive.get_signal(**additive.get_controls(c['amplitudes'], c['harmonic_distribution'],
                                                      c['inharm_coef'], c['f0_hz']))
    nz = noise.get_signal(**noise.get_controls(c['magnitudes']))
    assert sig.shape == nz.shape == (1, 30 * 96)
    assert pg.processors[0].sample_rate == 24000        # piano_model.py:70-72


@pytest.mark.parametrize('fast', [True, False])
def test_decompose_equals_the_reference_loop(fast):
    """synthesize_from_csv.py:92-120 (--decompose): the un-reverbed mix and the sums over the voices of the additive and the
    noise signals, which the reference gets by calling processors[:2] once per voice -- ProcessorGroup.decompose forms them
    on the batched route (compacted bank, the noise kernel's voice sums) and, node by node, with the reference's loop."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(21)
    B, P, T, H, K, sr = 3, 4, 50, 128, 96, 24000
    N = T * 96
    dag, _ = _build(dp, P, sr)
    pg = dp.ProcessorGroup(dag, fast_path=fast)
    feats = {}
    for i in range(P):
        for k, v in synth_controls(rng, B, T, H, K=K, silent_frac=0.0, midi_lo=40, midi_hi=90).items():
            feats[f'{k}_{i}'] = torch.as_tensor(v, device='cuda')
    feats['reverb_ir'] = torch.as_tensor(synth_ir(rng, B, 4800), device='cuda')
    noise = [torch.as_tensor(rng.uniform(-1, 1, [B, N]).astype(np.float32), device='cuda') for _ in range(P)]
    d = pg.decompose(feats, noise=noise)
    additive, noise_p = pg.processors[:2]
    a = z = None
    for i in range(P):                                     # the reference's loop
        x = additive.get_signal(**additive.get_controls(*[feats[f'{k}_{i}'] for k in ('amplitudes', 'harmonic_distribution',
                                                                                    'inharm_coef', 'f0_hz')]))
        y = noise_p.get_signal(**noise_p.get_controls(feats[f'magnitudes_{i}']), noise=noise[i])
        a = x if a is None else a + x
        z = y if z is None else z + y
    for k, ref in (('additive', a), ('noise', z), ('dry', a + z)):
        assert d[k].shape == (B, N)
        assert float((d[k] - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), k
    full = pg(feats, return_outputs_dict=True, noise=noise)
    assert float((d['signal'] - full['signal']).abs().max()) < 2e-5 * max(1.0, float(full['signal'].abs().max()))
    assert float((d['dry'] - full['controls']['add']['signal']).abs().max()) < 2e-5


@pytest.mark.parametrize('shape', ['default_model', 'surrogate'])
def test_decompose_on_the_other_node_lists(shape):
    """ProcessorGroup.decompose on default_model.py's node list (noise node first, explicit Add nodes) and on a
    SurrogateAdditive group: the sums over the voices against the stems of the every-stem route."""
    import bench
    import ddsp_piano_amd as dp
    dev = torch.device('cuda', 0)
    B, P, T, H, K, sr = 3, 5, 60, 96, 64, 16000
    N = T * 64
    feats, _ = bench.make_features(B, P, T, H, K, 1, 2400, dev, seed=77, silent_frac=0.2)
    if shape == 'surrogate':
        g = torch.Generator(device=dev)
        g.manual_seed(78)
        for i in range(P):
            feats[f'decays_{i}'] = 0.9985 + 0.0015 * torch.rand(B, T, H, generator=g, device=dev)
            feats[f'decay_time_{i}'] = (torch.arange(T, device=dev, dtype=torch.float32) % 17).view(1, T, 1).expand(B, T, 1).contiguous()
        pg = bench.build_shipped_group(dp, 'surrogate', P, sr)
    else:
        pg = bench.build_default_model_group(dp, P, sr)
    noise = 2.0 * torch.rand(B, P, N, device=dev) - 1.0
    d = pg.decompose(feats, noise=noise)
    stems = pg(feats, return_outputs_dict=True, need_stems=True, noise=noise)['controls']['voices']
    for k in ('additive', 'noise'):
        ref = stems[k].sum(dim=1)
        assert d[k].shape == (B, N) and float((d[k] - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), k
    full = pg(feats, return_outputs_dict=True, noise=noise)
    assert float((d['signal'] - full['signal']).abs().max()) < 2e-5 * max(1.0, float(full['signal'].abs().max()))
    assert float((d['dry'] - (d['additive'] + d['noise'])).abs().max()) < 2e-5 * max(1.0, float(d['dry'].abs().max()))


def test_long_segment_whole_file_mode():
    """synthesize_midi_file.py feeds the WHOLE file as one segment: many chunks of the
    angular cumsum (long float32 offset sums), many FIR frames, a multi-million point reverb FFT."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(77)
    B, P, T, H, K, S, sr, L = 1, 2, 2600, 64, 32, 1, 16000, 16000
    N = T * (sr // 250)                        # 166 400 samples = 167 chunks of 1000
    feats = _features(rng, B, P, T, H, K, S, L)
    for i in range(P):                         # a pitch change mid-file: frequencies move between frames
        f0 = feats[f'f0_hz_{i}']
        f0[:, T // 3:] *= np.float32(2 ** (3 / 12))
        f0[:, 2 * T // 3:] *= np.float32(2 ** (-5 / 12))
    noises = [rng.uniform(-1, 1, [B, N]).astype(np.float32) for _ in range(P)]
    odag, _ = _build(O, P, sr)
    ref = O.ProcessorGroup(odag)(feats, extra_kwargs={'noise': [{'noise': z} for z in noises]})
    gdag, gnoise = _build(dp, P, sr)
    got = dp.ProcessorGroup(gdag)({k: torch.as_tensor(v, device='cuda') for k, v in feats.items()},
                                  noise=[torch.as_tensor(z, device='cuda') for z in noises]).cpu().numpy()
    assert got.shape == ref.shape == (B, N)
    err = rms_err(got, ref)
    assert err < TOL * max(1.0, rms(ref)), f'{err:.3e} vs rms {rms(ref):.3e}'


# ddsp_piano/default_model.py:20-85 builds the group with explicit Add nodes (add_i, sub_add_i) and the
# noise synth first; since round 4 that shape runs on the batched route too (polyphonic._recognise_default_model).
def _default_model_dag(mod, P, sr, reverb_length=2000):
    noise = (mod.FilteredNoise if mod is O else mod.DynamicSizeFilteredNoise)(name='noise', frame_rate=250, sample_rate=sr)
    additive = mod.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True)
    dag = [(noise, ['magnitudes_0']),
           (additive, ['amplitudes_0', 'harmonic_distribution_0', 'inharm_coef_0', 'f0_hz_0']),
           (mod.Add(name='add_0'), ['noise/signal', 'additive/signal'])]
    for i in range(1, P):
        dag.append((additive, [f'amplitudes_{i}', f'harmonic_distribution_{i}', f'inharm_coef_{i}', f'f0_hz_{i}']))
        dag.append((noise, [f'magnitudes_{i}']))
        dag.append((mod.Add(name=f'sub_add_{i}'), ['noise/signal', 'additive/signal']))
        dag.append((mod.Add(name=f'add_{i}'), [f'add_{i - 1}/signal', f'sub_add_{i}/signal']))
    dag.append((mod.Reverb(trainable=False, reverb_length=reverb_length), [f'add_{P - 1}/signal', 'reverb_ir']))
    return dag, noise


def test_default_model_dag_shape():
    """The node list of default_model.py on the batched route (round 4): every need_stems level against the oracle's walk
    and against this package's own node-by-node walk."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import polyphonic
    rng = np.random.default_rng(31)
    B, P, T, H, K, S, sr, L = 2, 3, 30, 96, 64, 2, 16000, 2000
    N = T * 64
    feats = _features(rng, B, P, T, H, K, S, L)
    noises = [rng.uniform(-1, 1, [B, N]).astype(np.float32) for _ in range(P)]
    odag, _ = _default_model_dag(O, P, sr)
    ref = O.ProcessorGroup(odag)(feats, return_outputs_dict=True, extra_kwargs={'noise': [{'noise': z} for z in noises]})
    gfeats = {k: torch.as_tensor(v, device='cuda') for k, v in feats.items()}
    gnoises = [torch.as_tensor(z, device='cuda') for z in noises]
    gdag, gnoise = _default_model_dag(dp, P, sr)
    plan = polyphonic.recognise(gdag)
    assert plan is not None and plan.shape == 'default_model' and plan.n_synths == P
    assert [a.name for a in plan.adds] == ['add_0', 'add_1', 'add_2'] and [s.name for s in plan.subs[1:]] == ['sub_add_1', 'sub_add_2']
    pg = dp.ProcessorGroup(gdag)
    walk = dp.ProcessorGroup(gdag, fast_path=False)(gfeats, return_outputs_dict=True, noise=gnoises)
    # the reference's call form: the dictionary holds what run_dag leaves -- every `add_i` / `sub_add_i` (round 5: this
    # node list names every Add node, so the call form takes every voice's stems)
    def keys_of(d, prefix=''):
        ks = set()
        for k, v in d.items():
            if k == 'inputs' or k in gfeats:
                continue
            ks.add(prefix + k)
            if isinstance(v, dict):
                ks |= keys_of(v, prefix + k + '/')
        return ks
    whole = pg(gfeats, return_outputs_dict=True, noise=gnoises)
    assert keys_of(walk['controls']) <= keys_of(whole['controls']), keys_of(walk['controls']) - keys_of(whole['controls'])
    for i in range(P):
        for k in ([f'add_{i}/signal', f'add_{i}/controls/signal_one'] + ([f'sub_add_{i}/signal', f'sub_add_{i}/controls/signal_one'] if i else [])):
            a = whole['controls']
            b = ref['controls']
            for part in k.split('/'):
                a, b = a[part], b[part]
            assert rms_err(a.cpu().numpy(), b) < TOL, k
    assert rms_err(whole['signal'].cpu().numpy(), ref['signal']) < TOL * max(1.0, rms(ref['signal']))
    # the reduced dictionary (explicit opt-in): the last voice's pair, the mix before it and the dry mix
    out = pg(gfeats, return_outputs_dict=True, noise=gnoises, need_stems='last')
    assert pg._plan is plan or pg._plan.shape == 'default_model'
    assert rms_err(out['signal'].cpu().numpy(), ref['signal']) < TOL * max(1.0, rms(ref['signal']))
    ctl = out['controls']
    for k in (f'add_{P - 1}', f'sub_add_{P - 1}', f'add_{P - 2}', 'noise', 'additive'):
        assert rms_err(ctl[k]['signal'].cpu().numpy(), ref['controls'][k]['signal']) < TOL, k
        assert (ctl[k]['signal'] - walk['controls'][k]['signal']).abs().max().item() < 5e-6, k
    assert ctl[f'add_{P - 1}']['controls']['signal_one'] is ctl[f'add_{P - 2}']['signal']
    assert ctl[f'add_{P - 1}']['controls']['signal_two'] is ctl[f'sub_add_{P - 1}']['signal']
    assert ctl['out'] is ctl['reverb'] and 'voices' not in ctl           # (the compacted route: no per-voice stems)
    assert [p.name for p in pg.processors][:3] == ['noise', 'additive', 'add_0']
    # audio only
    audio = pg(gfeats, noise=gnoises)
    assert (audio - out['signal']).abs().max().item() < 5e-6
    # every stem: the whole dictionary of the reference's walk
    full = pg(gfeats, return_outputs_dict=True, noise=gnoises, need_stems=True)['controls']
    for i in range(P):
        for k in ([f'add_{i}'] + ([f'sub_add_{i}'] if i else [])):
            assert rms_err(full[k]['signal'].cpu().numpy(), ref['controls'][k]['signal']) < TOL, k
            assert rms_err(full[k]['controls']['signal_two'].cpu().numpy(), ref['controls'][k]['controls']['signal_two']) < TOL, k
    # a single voice (3 nodes + reverb) and no reverb at all
    for p1, with_reverb in ((1, True), (2, False)):
        d1, _ = _default_model_dag(dp, p1, sr)
        o1, _ = _default_model_dag(O, p1, sr)
        if not with_reverb:
            d1, o1 = d1[:-1], o1[:-1]
        assert polyphonic.recognise(d1).shape == 'default_model'
        r1 = O.ProcessorGroup(o1)(feats, return_outputs_dict=True, extra_kwargs={'noise': [{'noise': z} for z in noises[:p1]]})
        g1 = dp.ProcessorGroup(d1)(gfeats, return_outputs_dict=True, noise=gnoises[:p1])
        assert rms_err(g1['signal'].cpu().numpy(), r1['signal']) < TOL * max(1.0, rms(r1['signal']))
        assert rms_err(g1['controls'][f'add_{p1 - 1}']['signal'].cpu().numpy(), r1['controls'][f'add_{p1 - 1}']['signal']) < TOL


def test_default_model_dag_at_dafx22_dims():
    """default_model.py's own dimensions (16 kHz, two sub-strings, poly 16) at batch 64: the batched route against the node
    walk on every row and against the oracle on two."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(32)
    B, P, T, H, K, S, sr, L = 64, 16, 125, 96, 64, 2, 16000, 24000
    N = T * 64
    feats = _features(rng, B, P, T, H, K, S, L)
    noises = [rng.uniform(-1, 1, [B, N]).astype(np.float32) for _ in range(P)]
    gfeats = {k: torch.as_tensor(v, device='cuda') for k, v in feats.items()}
    gnoises = [torch.as_tensor(z, device='cuda') for z in noises]
    gdag, _ = _default_model_dag(dp, P, sr, reverb_length=L)
    out = dp.ProcessorGroup(gdag)(gfeats, return_outputs_dict=True, noise=gnoises)
    walk = dp.ProcessorGroup(gdag, fast_path=False)(gfeats, return_outputs_dict=True, noise=gnoises)
    scale = max(1.0, rms(walk['signal'].cpu().numpy()))
    assert rms_err(out['signal'].cpu().numpy(), walk['signal'].cpu().numpy()) < 2e-6 * scale
    for k in (f'add_{P - 1}', f'sub_add_{P - 1}', f'add_{P - 2}'):
        assert rms_err(out['controls'][k]['signal'].cpu().numpy(), walk['controls'][k]['signal'].cpu().numpy()) < 2e-6
    segs = [3, 40]
    sub = {k: v[segs] for k, v in feats.items()}
    odag, _ = _default_model_dag(O, P, sr, reverb_length=L)
    ref = O.ProcessorGroup(odag)(sub, return_outputs_dict=True, extra_kwargs={'noise': [{'noise': z[segs]} for z in noises]})
    assert rms_err(out['signal'][segs].cpu().numpy(), ref['signal']) < TOL * max(1.0, rms(ref['signal']))
    assert rms_err(out['controls'][f'add_{P - 1}']['signal'][segs].cpu().numpy(), ref['controls'][f'add_{P - 1}']['signal']) < TOL


def test_config5_shape_48k_poly32_long_ir():
    """BASELINE config 5 dims on a small batch: 48 kHz (U = 192), poly 32, H = 128, 10 s IR (2^20 FFT)."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(55)
    B, P, T, H, K, S, sr, L = 2, 32, 125, 128, 96, 1, 48000, 480000
    N = T * 192
    feats = {k: torch.as_tensor(v, device='cuda') for k, v in _features(rng, B, P, T, H, K, S, L).items()}
    dag, _ = _build(dp, P, sr)
    pg = dp.ProcessorGroup(dag)
    assert pg.additive.upsampling == 192
    out = pg(feats, return_outputs_dict=True)
    y = out['signal']
    assert y.shape == (B, N) and torch.isfinite(y).all()
    # the dry mix is what the reverb was fed; wet = dry + conv(dry, masked ir): check on a delta-like IR too
    dry = out['controls']['add']['signal']
    ir = torch.zeros(B, L, device='cuda')
    ir[:, 4321] = 0.25
    wet = dp.Reverb().get_signal(dry, ir)
    exp = dry.clone()
    exp[:, 4321:] += 0.25 * dry[:, :-4321]
    assert (wet - exp).abs().max().item() < 2e-5 * max(1.0, dry.abs().max().item())
    # batched route == node-by-node route (same noise)
    noise = torch.as_tensor(rng.uniform(-1, 1, [B, P, N]).astype(np.float32), device='cuda')
    daga, _ = _build(dp, P, sr, with_reverb=False)
    dagb, _ = _build(dp, P, sr, with_reverb=False)
    a, b = dp.ProcessorGroup(daga, fast_path=True), dp.ProcessorGroup(dagb, fast_path=False)
    assert (a(feats, noise=noise) - b(feats, noise=noise)).abs().max().item() < 5e-6


def test_monophonic_group_and_config1_dry():
    """BASELINE config 1 wiring: poly = 1, no reverb (dry) -- the smallest polyphonic DAG."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(2)
    B, P, T, H, K, S, sr = 1, 1, 250, 64, 64, 1, 24000
    feats = _features(rng, B, P, T, H, K, S, 100)
    noise = rng.uniform(-1, 1, [B, T * 96]).astype(np.float32)
    odag, _ = _build(O, P, sr, with_reverb=False)
    ref = O.ProcessorGroup(odag)(feats, extra_kwargs={'noise': [{'noise': noise}]})
    for fast in (True, False):
        gdag, gnoise = _build(dp, P, sr, with_reverb=False)
        got = dp.ProcessorGroup(gdag, fast_path=fast)({k: torch.as_tensor(v, device='cuda') for k, v in feats.items()},
                                                      noise=[torch.as_tensor(noise, device='cuda')])
        assert got.shape == (B, 24000) and rms_err(got.cpu().numpy(), ref) < TOL


def test_compacted_additive_equals_voice_stems():
    """ddspp_polyphonic_additive (lanes only for audible oscillators, per-segment mix) vs the sum of the
    per-voice stems of ddspp_harmonic_synthesis: same phases bit for bit, only the summation order differs."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(42)
    for (B, P, T, H, S, sr) in [(3, 16, 60, 128, 1, 24000), (2, 5, 40, 96, 2, 16000), (1, 32, 30, 64, 2, 16000)]:
        U = sr // 250
        N = T * U
        R = B * P
        raw = synth_controls(rng, R, T, H, S=S, silent_frac=0.3)
        syn = dp.MultiInharmonic(sample_rate=sr, inference=True)
        ctl = syn.get_controls(*[torch.as_tensor(raw[k], device='cuda') for k in
                                 ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')])
        amp = ctl['amplitudes'].reshape(R, T).contiguous()
        stems = core.harmonic_synthesis_fused(ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'],
                                              N, sr, True).reshape(B, P, N)
        for spans in (0, 1, 7):
            mix = core.polyphonic_additive(ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'],
                                           B, N, sr, spans=spans)
            assert mix.shape == (B, N)
            assert (mix - stems.sum(dim=1)).abs().max().item() < 3e-6, (B, P, H, S, spans)
            # the same rows in voice-major order [P, B]: identical lanes, identical arithmetic
            vm = [x.reshape((B, P) + x.shape[1:]).transpose(0, 1).reshape(x.shape).contiguous()
                  for x in (ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'])]
            mix_vm = core.polyphonic_additive(*vm, B, N, sr, spans=spans, voice_major=True)
            assert torch.equal(mix_vm, mix), (B, P, H, S, spans)
            # the per-frame audible-harmonic counts of get_controls replace the scan of [R, T, H]: same lanes, same bits
            raw_t = [torch.as_tensor(raw[k], device='cuda') for k in ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')]
            cnt = syn._controls(*raw_t, want_counts=True)['_audible']
            assert cnt.dtype == torch.int32 and cnt.shape == (R, T)
            prod = (ctl['amplitudes'] * ctl['harmonic_distribution']) != 0
            want = torch.where(prod.any(-1), H - prod.flip(-1).to(torch.int32).argmax(-1), torch.zeros_like(cnt))
            assert torch.equal(cnt & 0xffff, want.to(torch.int32))
            f0r, inr = raw_t[3], raw_t[2].clamp(min=0)
            moved = torch.zeros_like(cnt, dtype=torch.bool)
            moved[:, 1:] = (f0r[:, 1:] != f0r[:, :-1]).any(-1) | (inr[:, 1:, 0] != inr[:, :-1, 0])
            assert torch.equal((cnt >> 16) & 1, moved.to(torch.int32))     # bit 16: the frequencies may have moved
            mix_cnt = core.polyphonic_additive(ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'],
                                               B, N, sr, spans=spans, audible=cnt)
            assert torch.equal(mix_cnt, mix), (B, P, H, S, spans)
            # the last voice kept apart (what the outputs dictionary of the reference's DAG holds): its oscillators get
            # wavefront slots of their own, every oscillator still runs exactly once
            rest, last_v = core.polyphonic_additive(ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'],
                                                    B, N, sr, spans=spans, audible=cnt, split_last=True)
            assert (last_v - stems[:, P - 1]).abs().max().item() < 3e-6, (B, P, H, S, spans)
            assert (rest - stems[:, :P - 1].sum(dim=1)).abs().max().item() < 3e-6, (B, P, H, S, spans)
            rest_vm, last_vm = core.polyphonic_additive(*vm, B, N, sr, spans=spans, voice_major=True, split_last=True)
            assert torch.equal(rest_vm, rest) and torch.equal(last_vm, last_v)
            # harmonic_shifts formed inside the kernels from the raw inharm_coef (no [R, T, H] tensor): the same bits
            inh_raw = raw_t[2].reshape(R, T).contiguous()
            mix_inh = core.polyphonic_additive(ctl['f0_hz'], amp, ctl['harmonic_distribution'], None, B, N, sr, spans=spans,
                                               audible=cnt, inharm_coef=inh_raw)
            assert torch.equal(mix_inh, mix), (B, P, H, S, spans)
            no_shift = core.polyphonic_additive(ctl['f0_hz'], amp, ctl['harmonic_distribution'], None, B, N, sr, spans=spans)
            assert not torch.equal(no_shift, mix) or float(inh_raw.abs().max()) == 0.0
    # all voices silent: zeros
    z = torch.zeros(4, 20, 8, device='cuda')
    out = core.polyphonic_additive(torch.full((4, 20, 1), 100.0, device='cuda'), torch.zeros(4, 20, device='cuda'), z, z,
                                   2, 20 * 96, 24000)
    assert out.shape == (2, 1920) and (out == 0).all()


def test_paired_substrings_in_the_compacted_bank(monkeypatch):
    """Round 5: with two sub-strings (dafx22 configs, default_model.py) a lane of the compacted bank carries BOTH sub-strings of
    one (voice, harmonic) and evaluates one Hann cross-fade for the pair, a (cos0 + cos1) -- MultiInharmonic shares amplitudes,
    distribution and shifts between them (inharm_synth.py:279-292).  Against the unpaired kernel (DDSPP_OSC_PAIR=0), the
    per-voice stems and the oracle; held notes, moving notes, and a partial that lies in the detune gap around Nyquist (one
    sub-string audible, the other masked by remove_above_nyquist: the pair falls back to separate amplitudes)."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(52)
    set_option(monkeypatch, 'DDSPP_OSC_COMPACT_VPL1_BELOW', '0')          # two oscillators per lane even for these few rows
    keys = ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')
    for (B, P, T, H, sr, moving) in [(2, 5, 40, 96, 16000, False), (3, 16, 30, 128, 24000, True), (1, 32, 30, 64, 16000, False)]:
        U, S = sr // 250, 2
        N, R = T * U, B * P
        raw = synth_controls(rng, R, T, H, S=S, silent_frac=0.2, midi_lo=40 if moving else 21, midi_hi=100)
        if moving:      # vibrato + a glide: partials cross Nyquist inside frames, the two sub-strings at different samples
            tt = np.arange(T, dtype=np.float32)[None, :, None]
            raw['f0_hz'] = (raw['f0_hz'] * (1 + 0.004 * np.sin(0.13 * tt + rng.uniform(0, 6, [R, 1, 1])) - 0.0015 * tt)).astype(np.float32)
        syn = dp.MultiInharmonic(sample_rate=sr, inference=True)
        dev = [torch.as_tensor(raw[k], device='cuda') for k in keys]
        ctl = syn._controls(*dev, want_counts=True)
        amp = ctl['amplitudes'].reshape(R, T).contiguous()
        args = (ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, sr)
        stems = core.harmonic_synthesis_fused(*args[:4], N, sr, True).reshape(B, P, N)
        for spans in (0, 1, 5):
            set_option(monkeypatch, 'DDSPP_OSC_PAIR', '0')
            plain = core.polyphonic_additive(*args, spans=spans, audible=ctl['_audible'])
            set_option(monkeypatch, 'DDSPP_OSC_PAIR')
            mix = core.polyphonic_additive(*args, spans=spans, audible=ctl['_audible'])
            scale = max(1.0, float(plain.abs().max()))
            assert (mix - plain).abs().max().item() < 2e-6 * scale, (B, P, H, spans)       # a c0 + a c1 against a (c0 + c1)
            assert (mix - stems.sum(dim=1)).abs().max().item() < 4e-6 * scale, (B, P, H, spans)
            rest, last_v = core.polyphonic_additive(*args, spans=spans, audible=ctl['_audible'], split_last=True)
            assert (last_v - stems[:, P - 1]).abs().max().item() < 3e-6 * scale
            assert (rest - stems[:, :P - 1].sum(dim=1)).abs().max().item() < 4e-6 * scale
            vm = [x.reshape((B, P) + x.shape[1:]).transpose(0, 1).reshape(x.shape).contiguous() for x in args[:4]]
            cnt_vm = ctl['_audible'].reshape(B, P, T).transpose(0, 1).reshape(R, T).contiguous()
            assert torch.equal(core.polyphonic_additive(*vm, B, N, sr, spans=spans, voice_major=True, audible=cnt_vm), mix)
            inh_raw = dev[2].reshape(R, T).contiguous()                                       # shifts formed in the kernel
            mix_inh = core.polyphonic_additive(ctl['f0_hz'], amp, ctl['harmonic_distribution'], None, B, N, sr, spans=spans,
                                               audible=ctl['_audible'], inharm_coef=inh_raw)
            assert torch.equal(mix_inh, mix), (B, P, H, spans)
        osyn = O.MultiInharmonic(sample_rate=sr, inference=True)
        ref = osyn(*[raw[k][:P] for k in keys]).sum(0)                                         # segment 0 against the oracle
        assert rms_err(mix[0].cpu().numpy(), ref) < TOL * max(1.0, rms(ref)), (B, P, H)
    # the detune gap: harmonic 12 of sub-string 0 at 11 999.88 Hz (audible), of sub-string 1 at 12 000 Hz = Nyquist (masked
    # by cos_oscillator_bank, inharm_synth.py:65-67; get_controls cuts by sub-string 0 only, :185-187)
    B, P, T, H, sr = 4, 2, 25, 16, 24000
    R, N = B * P, T * 96
    raw = synth_controls(rng, R, T, H, S=2, silent_frac=0.0)
    raw['f0_hz'][..., 0], raw['f0_hz'][..., 1] = np.float32(999.99), np.float32(1000.0)
    raw['f0_hz'][1::2] *= np.float32(0.5)                                                      # every other voice: far from the gap
    raw['inharm_coef'][:] = 0.0
    syn = dp.MultiInharmonic(sample_rate=sr, inference=True)
    dev = [torch.as_tensor(raw[k], device='cuda') for k in keys]
    ctl = syn._controls(*dev, want_counts=True)
    assert int((ctl['_audible'][0] & 0xffff).max()) == 12                                      # harmonic 12 is the last one kept
    amp = ctl['amplitudes'].reshape(R, T).contiguous()
    mix = core.polyphonic_additive(ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, sr,
                                   audible=ctl['_audible'])
    osyn = O.MultiInharmonic(sample_rate=sr, inference=True)
    for b in range(B):
        ref = osyn(*[raw[k][b * P:(b + 1) * P] for k in keys]).sum(0)
        assert rms_err(mix[b].cpu().numpy(), ref) < TOL * max(1.0, rms(ref)), b
    # the unpaired kernel (one oscillator per lane, each masked by its own frequency) agrees ...
    set_option(monkeypatch, 'DDSPP_OSC_PAIR', '0')
    plain = core.polyphonic_additive(ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, sr,
                                     audible=ctl['_audible'])
    assert (mix - plain).abs().max().item() < 2e-6 * max(1.0, float(plain.abs().max()))
    set_option(monkeypatch, 'DDSPP_OSC_PAIR')
    # ... and the gap is really there: with sub-string 1 moved to sub-string 0's frequency its harmonic 12 is below
    # Nyquist and sounds, so the paired kernel must have silenced exactly that partial above -- the two renders differ by about
    # its amplitude, amp * hd[11] / 2, in the voices at 1 kHz (and not at all in the voices an octave lower)
    f0_same = ctl['f0_hz'].clone()
    f0_same[..., 1] = f0_same[..., 0]
    unmasked = core.polyphonic_additive(f0_same, amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, sr,
                                        audible=ctl['_audible'])
    partial = float((amp[0::2, :, None] * ctl['harmonic_distribution'][0::2, :, 11:12]).abs().max()) / 2
    assert partial > 1e-5
    assert (unmasked - mix).abs().max().item() > 0.5 * partial


def test_moving_frequencies_prepass_parts_and_nyquist_crossings(monkeypatch):
    """Every frame's frequencies move (vibrato + glide, some partials gliding through Nyquist): the memoised pre-pass
    scans every chunk sample by sample, sections of four wavefronts per (row, 64 oscillators) sharing the chunks.  Same start
    phases bit for bit as one wavefront walking the whole row (whatever the section length), and the audio matches the oracle -- also with
    normalize_below_nyquist=False, where a partial above Nyquist keeps its amplitude in the controls and only the
    sample-rate mask of cos_oscillator_bank (inharm_synth.py:65-67) silences it."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(41)
    B, P, T, H, sr = 5, 4, 120, 128, 24000            # R * 2 groups >= ... the batch form needs R >= 256 rows for the memo
    B = 64                                            # 256 rows: the memoised pre-pass is the one that runs
    N = T * 96
    R = B * P
    raw = synth_controls(rng, R, T, H, S=1, silent_frac=0.1, midi_lo=60, midi_hi=100)
    tt = np.arange(T, dtype=np.float32)[None, :, None]
    raw['f0_hz'] = (raw['f0_hz'] * (1 + 0.004 * np.sin(0.13 * tt + rng.uniform(0, 6, [R, 1, 1])) - 0.0015 * tt)
                    ).astype(np.float32)              # downward glide: high partials come back below Nyquist
    for nbn in (True, False):
        syn = dp.MultiInharmonic(sample_rate=sr, inference=True, normalize_below_nyquist=nbn)
        osyn = O.MultiInharmonic(sample_rate=sr, inference=True, normalize_below_nyquist=nbn)
        dev = [torch.as_tensor(raw[k], device='cuda') for k in ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')]
        ctl = syn._controls(*dev, want_counts=True)
        amp = ctl['amplitudes'].reshape(R, T).contiguous()
        args = (ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, sr)
        mix = core.polyphonic_additive(*args, audible=ctl['_audible'])
        set_option(monkeypatch, 'DDSPP_OSC_PREPASS_ONE_WAVE', '1')
        one = core.polyphonic_additive(*args, audible=ctl['_audible'])
        set_option(monkeypatch, 'DDSPP_OSC_PREPASS_ONE_WAVE')
        assert torch.equal(mix, one), nbn
        for run in ('1', '2'):                         # 11 chunks in sections of 4 / 8: three and two workgroups per (row, group)
            set_option(monkeypatch, 'DDSPP_OSC_PREPASS_RUN', run)
            assert torch.equal(core.polyphonic_additive(*args, audible=ctl['_audible']), one), (nbn, run)
        set_option(monkeypatch, 'DDSPP_OSC_PREPASS_RUN')
        stems = core.harmonic_synthesis_fused(*args[:4], N, sr, True).reshape(B, P, N)
        assert (mix - stems.sum(dim=1)).abs().max().item() < 5e-6, nbn
        rows = [0, 7, R - 1]                          # the oracle on a few voice rows
        ref = osyn(*[raw[k][rows] for k in ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')])
        got = stems.reshape(R, N)[rows].cpu().numpy()
        assert rms_err(got, ref) < TOL, nbn
        one_seg = core.polyphonic_additive(*[a[:P].contiguous() for a in args[:4]], 1, N, sr)     # compact kernel, 1 segment
        ref_seg = osyn(*[raw[k][:P] for k in ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')]).sum(0)
        assert rms_err(one_seg.cpu().numpy()[0], ref_seg) < TOL, nbn


def test_compacted_scan_of_moving_chunks(monkeypatch):
    """Round 5: the chunks in which a segment's frequencies move are scanned by bank_scan_kernel with the bank's packing
    (lanes for the harmonics below each row's audible maximum, voices back to back) instead of one wavefront per (row, 64
    harmonics); held chunks stay with the memoised pre-pass.  Same span starts, bit for bit, as the pre-pass alone
    (DDSPP_OSC_COMPACT_SCAN=0): batches that mix held, moving and silent voices, notes that move only for a part of the
    segment, two sub-strings, voice-major rows, the last voice kept apart."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(77)
    keys = ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')
    # (oscillators per row: 128, 192 = three 64-groups, 64, and 96 -- not a multiple of 64: whole rows of two per lane)
    for (B, P, T, H, S, sr) in [(64, 4, 120, 128, 1, 24000), (32, 8, 150, 96, 2, 16000), (16, 16, 125, 64, 1, 24000),
                                (32, 8, 140, 96, 1, 16000)]:
        U = sr // 250
        N, R = T * U, B * P
        raw = synth_controls(rng, R, T, H, S=S, silent_frac=0.15, midi_lo=30, midi_hi=100)
        tt = np.arange(T, dtype=np.float32)[None, :, None]
        kind = rng.integers(0, 3, size=[R, 1, 1])                 # 0: held, 1: vibrato all along, 2: a glide in the middle third only
        vib = 1 + 0.004 * np.sin(0.13 * tt + rng.uniform(0, 6, [R, 1, 1]))
        glide = 1 + 0.02 * np.clip((tt - T / 3) / (T / 3), 0, 1)
        raw['f0_hz'] = (raw['f0_hz'] * np.where(kind == 1, vib, np.where(kind == 2, glide, 1.0))).astype(np.float32)
        syn = dp.MultiInharmonic(sample_rate=sr, inference=True)
        dev = [torch.as_tensor(raw[k], device='cuda') for k in keys]
        ctl = syn._controls(*dev, want_counts=True, want_shifts=False)
        amp = ctl['amplitudes'].reshape(R, T).contiguous()
        inh = ctl['_inharm_coef'].reshape(R, T)
        kw = dict(audible=ctl['_audible'], inharm_coef=inh)
        args = (ctl['f0_hz'], amp, ctl['harmonic_distribution'], None, B, N, sr)
        set_option(monkeypatch, 'DDSPP_OSC_COMPACT_SCAN', '0')
        want = core.polyphonic_additive(*args, **kw)
        want_rest, want_last = core.polyphonic_additive(*args, split_last=True, **kw)
        set_option(monkeypatch, 'DDSPP_OSC_COMPACT_SCAN')
        got = core.polyphonic_additive(*args, **kw)
        assert torch.equal(got, want), (B, P, T, H, S)
        rest, last_v = core.polyphonic_additive(*args, split_last=True, **kw)
        assert torch.equal(rest, want_rest) and torch.equal(last_v, want_last), (B, P, T, H, S)
        vm = [x.reshape((B, P) + x.shape[1:]).transpose(0, 1).reshape(x.shape).contiguous()
              for x in (ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['_audible'], inh)]
        got_vm = core.polyphonic_additive(vm[0], vm[1], vm[2], None, B, N, sr, voice_major=True, audible=vm[3], inharm_coef=vm[4])
        assert torch.equal(got_vm, got), (B, P, T, H, S)
        # and against the oracle on one segment
        osyn = O.MultiInharmonic(sample_rate=sr, inference=True)
        ref = osyn(*[raw[k][:P] for k in keys]).sum(0)
        assert rms_err(got[0].cpu().numpy(), ref) < TOL * max(1.0, rms(ref)), (B, P, T, H, S)


def test_every_voice_stem_through_the_compacted_bank(monkeypatch):
    """Round 5: need_stems=True renders the additive stems with the compacted bank -- ddspp_polyphonic_stems: the voices of
    a segment packed into the same wavefronts in whole blocks of 32 oscillators, the harmonic sum stopped at voice
    boundaries (or, DDSPP_STEMS_SINGLE=1, ddspp_polyphonic_additive with every voice a segment of its own) -- instead of
    the per-voice fused kernel (DDSPP_NO_STEMS_COMPACT=1).  Same dictionary, same stems to float32 rounding of the harmonic
    sum, one and two sub-strings, rows voice major and segment major, harmonic counts that are no multiple of 32, against
    the oracle."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import _lib
    rng = np.random.default_rng(91)
    for (B, P, T, H, K, S, sr, L, vm) in [(3, 5, 60, 128, 96, 1, 24000, 3000, False), (2, 4, 50, 96, 64, 2, 16000, 2000, True),
                                          (2, 3, 40, 48, 32, 1, 8000, 1000, False), (3, 7, 50, 100, 64, 1, 16000, 1000, True),
                                          (64, 4, 125, 128, 96, 1, 24000, 3000, False), (40, 16, 130, 128, 96, 1, 24000, 3000, True)]:
        N = T * (sr // 250)
        feats = _features(rng, B, P, T, H, K, S, L)
        if B >= 40:                          # a vibrato on every other voice: the compacted scan of moving chunks
            tt = np.arange(T, dtype=np.float32)[None, :, None]
            for i in range(0, P, 2):
                feats[f'f0_hz_{i}'] = (feats[f'f0_hz_{i}'] * (1 + 0.004 * np.sin(0.13 * tt + i))).astype(np.float32)
        noises = [rng.uniform(-1, 1, [B, N]).astype(np.float32) for _ in range(P)]
        dev = {}
        for k in ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz', 'magnitudes'):
            whole = torch.as_tensor(np.stack([feats[f'{k}_{i}'] for i in range(P)], axis=0 if vm else 1), device='cuda')
            for i in range(P):
                dev[f'{k}_{i}'] = whole[i] if vm else whole[:, i]
        dev['reverb_ir'] = torch.as_tensor(feats['reverb_ir'], device='cuda')
        gdag, _ = _build(dp, P, sr)
        pg = dp.ProcessorGroup(gdag)
        nz = [torch.as_tensor(z, device='cuda') for z in noises]
        monkeypatch.setattr(_lib.options, 'no_stems_compact', True)
        want = pg(dev, return_outputs_dict=True, need_stems=True, noise=nz)
        monkeypatch.setattr(_lib.options, 'no_stems_compact', False)
        monkeypatch.setattr(_lib.options, 'stems_single', True)      # every voice a segment of its own (ddspp_polyphonic_additive, P' = 1)
        single = pg(dev, return_outputs_dict=True, need_stems=True, noise=nz)['controls']['voices']['additive']
        monkeypatch.setattr(_lib.options, 'stems_single', False)     # the default: ddspp_polyphonic_stems
        got = pg(dev, return_outputs_dict=True, need_stems=True, noise=nz)
        gv, wv = got['controls']['voices'], want['controls']['voices']
        assert float((single - wv['additive']).abs().max()) < 4e-6 * max(1.0, float(wv['additive'].abs().max())), (B, P, S)
        assert gv['additive'].shape == wv['additive'].shape == (B, P, N)
        scale = max(1.0, float(wv['additive'].abs().max()))
        assert float((gv['additive'] - wv['additive']).abs().max()) < 4e-6 * scale, (B, P, S)
        assert torch.equal(gv['noise'], wv['noise'])
        assert float((got['signal'] - want['signal']).abs().max()) < 2e-5 * max(1.0, float(want['signal'].abs().max()))
        for name in ('additive', 'noise', 'add', 'reverb'):
            assert set(got['controls'][name]['controls']) == set(want['controls'][name]['controls']), name
        for k, v in want['controls']['additive']['controls'].items():
            assert torch.equal(got['controls']['additive']['controls'][k], v), k
        assert torch.equal(got['controls']['noise']['controls']['magnitudes'], want['controls']['noise']['controls']['magnitudes'])
        if B < 40:
            osyn = O.MultiInharmonic(sample_rate=sr, inference=True)
            for i in (0, P - 1):
                ref = osyn(*[feats[f'{k}_{i}'] for k in ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')])
                assert rms_err(gv['additive'][:, i].cpu().numpy(), ref) < TOL * max(1.0, rms(ref)), (B, P, S, i)


def test_parallelizer_views_feed_the_group_without_copies():
    """The reference hands the group `features[k + '_i'] = features[k][i]` of the merged [P * B, T, C] control
    tensors (sub_modules.py:584-592): those views are consumed in place and give the same audio as separately
    allocated per-voice tensors."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import polyphonic
    rng = np.random.default_rng(21)
    B, P, T, H, K, sr, L = 2, 5, 30, 128, 96, 24000, 2000
    per_voice = [synth_controls(rng, B, T, H, S=1, K=K) for _ in range(P)]
    merged = {k: torch.as_tensor(np.ascontiguousarray(np.concatenate([c[k] for c in per_voice], axis=0)), device='cuda')
              for k in ('f0_hz', 'inharm_coef', 'amplitudes', 'harmonic_distribution', 'magnitudes')}
    par = dp.Parallelizer(n_synths=P)
    par.batch_size = B
    feats = par(dict(merged), parallelize=False)
    feats['reverb_ir'] = torch.as_tensor(synth_ir(rng, B, L), device='cuda')
    rows, vm = polyphonic._stack_voices([feats[f'harmonic_distribution_{i}'] for i in range(P)])
    assert vm is True and rows.data_ptr() == merged['harmonic_distribution'].data_ptr()
    separate = {k: (v.clone() if k[-1].isdigit() else v) for k, v in feats.items() if k not in merged}
    outs = []
    for f in (feats, separate):
        dag, noise = _build(dp, P, sr)
        noise.seed = 11
        outs.append(dp.ProcessorGroup(dag)({k: v for k, v in f.items() if k not in merged}))
    assert outs[0].shape == (B, T * 96) and torch.equal(outs[0], outs[1])


def test_outputs_dict_routes_agree():
    """group(features, return_outputs_dict=True) -- PianoModel.call's form (piano_model.py:160) -- runs the compacted
    mix and adds the last voice's stems; need_stems=True computes every voice's stems.  Same dict entries, same audio."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(8)
    B, P, T, H, K, S, sr, L = 2, 4, 25, 128, 96, 1, 24000, 3000
    feats = {k: torch.as_tensor(v, device='cuda') for k, v in _features(rng, B, P, T, H, K, S, L).items()}
    N = T * 96
    noises = [torch.as_tensor(rng.uniform(-1, 1, [B, N]).astype(np.float32), device='cuda') for _ in range(P)]
    outs = {}
    for mode in ('last', True, False):
        dag, gnoise = _build(dp, P, sr)
        pg = dp.ProcessorGroup(dag)
        outs[mode] = pg(feats, return_outputs_dict=True, need_stems=mode, noise=list(noises)) if mode is not False else \
            {'signal': pg(feats, noise=list(noises)), 'controls': None}
    last, full, audio = outs['last'], outs[True], outs[False]
    # same kernels as the audio-only call (which adds the noise of four voices at a time inside the noise kernel)
    assert (last['signal'] - audio['signal']).abs().max().item() < 5e-6
    assert (last['signal'] - full['signal']).abs().max().item() < 5e-6       # summation order of the voices differs
    cl, cf = last['controls'], full['controls']
    assert 'voices' not in cl and cf['voices']['additive'].shape == (B, P, N)
    for name in ('additive', 'noise'):
        # the last voice's stems: slots of its own in the compacted bank / a row of its own out of the voice-summing
        # noise kernel, against the per-voice kernels (same per-oscillator arithmetic, another summation order)
        assert (cl[name]['signal'] - cf[name]['signal']).abs().max().item() < 3e-6, name
        for k in cf[name]['controls']:
            assert torch.equal(cl[name]['controls'][k], cf[name]['controls'][k]), (name, k)
    assert set(cl['add']['controls']) == set(cf['add']['controls']) == {'signal_0', 'signal_1', 'signal_2'}
    for k in ('signal_0', 'signal_1', 'signal_2'):       # the last add node's operands: running mix, noise, additive
        assert (cl['add']['controls'][k] - cf['add']['controls'][k]).abs().max().item() < 5e-6, k
    last_step = (cl['add']['controls']['signal_0'] + cl['add']['controls']['signal_1']) + cl['add']['controls']['signal_2']
    assert torch.equal(last_step, cl['add']['signal'])    # the chain's last step is evaluated as the DAG writes it
    assert (cl['add']['signal'] - cf['add']['signal']).abs().max().item() < 5e-6
    assert cl['out'] is cl['reverb'] and 'reverb_ir' in cl and 'amplitudes_0' in cl


def test_chunk_prepass_kernel_equals_the_block_machinery(monkeypatch):
    """Few long rows: the dedicated chunk pre-pass (constant-frequency chunks as plain adds, moving ones interpolated
    out of LDS) and the tiled offset scan give the bits of osc_kernel<MODE_PREPASS> + the per-thread scan."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(31)
    B, P, T, H, sr = 1, 3, 3200, 128, 24000            # 307 chunks: tiled scan; pitch moves in the middle
    N = T * 96
    raw = synth_controls(rng, B * P, T, H, S=1, silent_frac=0.0)
    raw['f0_hz'][:, T // 2:] *= np.float32(2 ** (2 / 12))
    raw['f0_hz'][1, 100:160] *= np.linspace(1.0, 1.1, 60, dtype=np.float32)[:, None]        # a glide: frames differ
    syn = dp.MultiInharmonic(sample_rate=sr, inference=True)
    ctl = syn.get_controls(*[torch.as_tensor(raw[k], device='cuda') for k in
                             ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')])
    amp = ctl['amplitudes'].reshape(B * P, T).contiguous()
    args = (ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, sr)
    new = core.polyphonic_additive(*args)
    set_option(monkeypatch, 'DDSPP_OSC_OLD_CHUNK_PREPASS', '1')
    old = core.polyphonic_additive(*args)
    set_option(monkeypatch, 'DDSPP_OSC_OLD_CHUNK_PREPASS')
    assert torch.isfinite(new).all() and new.abs().max().item() > 0
    assert torch.equal(new, old)
    set_option(monkeypatch, 'DDSPP_OSC_SHORT_SCAN', '1')            # a thread per chain instead of the tiled scan
    assert torch.equal(core.polyphonic_additive(*args), new)
    set_option(monkeypatch, 'DDSPP_OSC_SHORT_SCAN')
    for spans in (1, 5, 307):                                 # and the span split stays invisible
        assert (core.polyphonic_additive(*args, spans=spans) - new).abs().max().item() < 3e-6


def test_long_rows_take_the_sectioned_memo_prepass(monkeypatch):
    """A file as one segment (few rows, thousands of frames): the memoised pre-pass in sections + the tiled scan with the
    per-row audible maximum from its own kernel (rows longer than 1 536 frames) and silent 64-oscillator groups left out
    -- the bits of the chunk-parallel pre-pass, which scans every chunk of every group."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(77)
    B, P, T, H, sr = 1, 16, 3300, 128, 24000           # 316 chunks; 16 rows x 2 groups x 14 sections
    N = T * 96
    raw = synth_controls(rng, B * P, T, H, S=1, silent_frac=0.25, midi_lo=30, midi_hi=100)
    raw['f0_hz'][:, T // 3:] *= np.float32(2 ** (-3 / 12))                                   # every voice changes note
    raw['f0_hz'][2, 500:700] *= np.linspace(1.0, 1.06, 200, dtype=np.float32)[:, None]       # a glide: moving chunks
    syn = dp.MultiInharmonic(sample_rate=sr, inference=True)
    ctl = syn._controls(*[torch.as_tensor(raw[k], device='cuda') for k in
                          ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')], want_counts=True)
    amp = ctl['amplitudes'].reshape(B * P, T).contiguous()
    args = (ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, sr)
    new = core.polyphonic_additive(*args, audible=ctl['_audible'])
    assert torch.isfinite(new).all() and new.abs().max().item() > 0
    set_option(monkeypatch, 'DDSPP_OSC_PLAIN_PREPASS', '1')
    assert torch.equal(core.polyphonic_additive(*args, audible=ctl['_audible']), new)
    set_option(monkeypatch, 'DDSPP_OSC_PLAIN_PREPASS')
    for run in ('2', '40'):                               # other section lengths (40: one section per (row, group))
        set_option(monkeypatch, 'DDSPP_OSC_PREPASS_RUN', run)
        assert torch.equal(core.polyphonic_additive(*args, audible=ctl['_audible']), new), run
    set_option(monkeypatch, 'DDSPP_OSC_PREPASS_RUN')
    stems = core.harmonic_synthesis_fused(*args[:4], N, sr, True).reshape(B, P, N)
    assert (new - stems.sum(1)).abs().max().item() < 6e-6 * max(1.0, float(new.abs().max()))


@pytest.mark.parametrize('B,P,T,H,K,L', [
    (2, 2, 1, 64, 32, 100),       # a single control frame
    (1, 1, 2, 64, 32, 7),         # two frames, one voice, a 7-tap "room"
    (1, 2, 5, 1, 32, 50),         # one harmonic
    (1, 2, 5, 200, 64, 50),       # a harmonic count that is no multiple of the wavefront
    (1, 17, 4, 32, 32, 50),       # more voices than the maestro model has
])
def test_odd_shapes_match_the_oracle(B, P, T, H, K, L):
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(100 + T + H + P)
    sr = 24000
    feats = _features(rng, B, P, T, H, K, 1, L)
    N = T * 96
    noises = [rng.uniform(-1, 1, [B, N]).astype(np.float32) for _ in range(P)]
    odag, _ = _build(O, P, sr)
    ref = O.ProcessorGroup(odag)(feats, extra_kwargs={'noise': [{'noise': z} for z in noises]})
    gdag, gnoise = _build(dp, P, sr)
    got = dp.ProcessorGroup(gdag)({k: torch.as_tensor(v, device='cuda') for k, v in feats.items()},
                                  noise=[torch.as_tensor(z, device='cuda') for z in noises]).cpu().numpy()
    assert got.shape == ref.shape == (B, N)
    assert rms_err(got, ref) < TOL * max(1.0, rms(ref))


def test_side_stream_route_gives_the_same_audio(monkeypatch):
    """Large batches enqueue the noise branch on a side stream (it overlaps the additive chain); same kernels, same
    arithmetic, so the audio is bit-identical to the single-stream order -- also when the outputs dict is requested."""
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(12)
    B, P, T, H, K, S, sr, L = 3, 4, 40, 128, 96, 1, 24000, 3000
    feats = {k: torch.as_tensor(v, device='cuda') for k, v in _features(rng, B, P, T, H, K, S, L).items()}
    outs = []
    for side in (False, True):
        set_option(monkeypatch, 'DDSPP_SIDE_STREAM', '1' if side else '0')
        set_option(monkeypatch, 'DDSPP_SIDE_STREAM_MIN', '1')
        for _ in range(3):                                   # a few calls in a row: buffers are recycled across streams
            dag, gnoise = _build(dp, P, sr)
            gnoise.seed = 5
            pg = dp.ProcessorGroup(dag)
            audio = pg(feats)
            full = pg(feats, return_outputs_dict=True)
        torch.cuda.synchronize()
        outs.append((audio.clone(), full['signal'].clone(), full['controls']['noise']['signal'].clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_noise_voice_sums_equal_the_per_voice_rows(monkeypatch):
    """Audio-only route: the fused noise kernel adds the filtered noise of four voices in registers.  Against the
    per-voice rows: the same numbers added in the same order, so the dry mix only moves by the final summation order."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(77)
    for vm in (False, True):
        B, P, T, K, U = 3, 8, 33, 96, 96
        N = T * U
        raw = torch.as_tensor(rng.normal(0, 2, [B * P, T, K]).astype(np.float32), device='cuda')
        x = torch.as_tensor(rng.uniform(-1, 1, [B * P, N]).astype(np.float32), device='cuda')
        syn = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=24000)
        rows = core.frequency_filter(x, raw, window_size=syn.window_size, raw_scale=syn.raw_scale())      # [R, N]
        sums = core.frequency_filter_voice_sums(x, raw, syn.window_size, syn.raw_scale(), P, 4, vm)         # [B * 2, N]
        assert sums.shape == (B * P // 4, N)
        per = rows.reshape(P, B, N).transpose(0, 1) if vm else rows.reshape(B, P, N)
        want = per.reshape(B, P // 4, 4, N)
        want = ((want[:, :, 0] + want[:, :, 1]) + want[:, :, 2]) + want[:, :, 3]
        assert torch.equal(sums.reshape(B, P // 4, N), want), vm
    # and through the group: same audio within the summation-order tolerance
    feats = {k: torch.as_tensor(v, device='cuda') for k, v in _features(rng, 2, 8, 30, 128, 96, 1, 2000).items()}
    outs = []
    for off in ('0', '1'):
        set_option(monkeypatch, 'DDSPP_NO_VOICE_SUMS', off)
        dag, gnoise = _build(dp, 8, 24000)
        gnoise.seed = 9
        outs.append(dp.ProcessorGroup(dag)(feats))
    assert (outs[0] - outs[1]).abs().max().item() < 5e-6


def test_sparse_get_controls_feeds_the_bank_the_same_bits(monkeypatch):
    """Round 6: on the compacted routes get_controls writes a frame's normalised harmonic_distribution only below the
    frame's audible count, in whole groups of 16 (ddspp_inharmonic_controls_sparse); the bank takes everything at or above
    the count as silent from the count.  Here the output buffer is POISONED with NaN first: what the kernel leaves out stays
    NaN, what it writes equals the dense kernel's, the last voice's rows are whole, and the bank (mix, split mix, every
    voice's stems; one and two sub-strings, both row orders) gives the dense route's audio bit for bit -- a NaN that reached
    a lane would show.  Then the whole group with the switch DDSPP_CONTROLS_DENSE_HD=1 against the default."""
    import ctypes
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import _lib, core
    lib = _lib.load()
    rng = np.random.default_rng(606)
    for (B, P, T, H, S, sr, vm) in [(3, 16, 60, 128, 1, 24000, False), (2, 5, 40, 96, 2, 16000, True),
                                    (4, 3, 50, 192, 1, 32000, False), (2, 6, 30, 48, 1, 8000, True)]:
        U = sr // 250
        N = T * U
        R = B * P
        raw = synth_controls(rng, R, T, H, S=S, silent_frac=0.3)
        raw_t = [torch.as_tensor(raw[k], device='cuda').contiguous()       # (raw pointers below: numpy leaves the smoothed hd strided)
                 for k in ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')]
        syn = dp.MultiInharmonic(sample_rate=sr, inference=True)
        dense = syn._controls(*raw_t, want_counts=True, want_shifts=False, last_voice_of=(P, vm))
        prm = core.scale_kind(syn.scale_fn)[1]
        amp_s = torch.empty_like(dense['amplitudes'])
        hd_s = torch.full_like(dense['harmonic_distribution'], float('nan'))
        shl = torch.empty_like(dense['_shifts_last'])
        cnt_s = torch.empty_like(dense['_audible'])
        _lib.check(lib.ddspp_inharmonic_controls_sparse(
            *[ctypes.c_void_p(x.data_ptr()) for x in raw_t], ctypes.c_void_p(amp_s.data_ptr()), ctypes.c_void_p(hd_s.data_ptr()),
            ctypes.c_void_p(shl.data_ptr()), ctypes.c_void_p(cnt_s.data_ptr()), R, T, H, S, P, int(vm), float(sr),
            float(syn.min_frequency), core.scale_kind(syn.scale_fn)[0], prm['exponent'], prm['max_value'], prm['threshold'],
            prm['gain'], 1, 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        assert torch.equal(cnt_s, dense['_audible']) and torch.equal(amp_s, dense['amplitudes'])
        assert torch.equal(shl, dense['_shifts_last'])
        cnt = (cnt_s & 0xffff).long()
        keep = ((cnt + 15) // 16 * 16)[..., None]                          # a row is written in whole groups of 16
        k = torch.arange(H, device='cuda')[None, None, :]
        rows = torch.arange(R, device='cuda')
        is_last = (rows >= (P - 1) * B) if vm else (rows % P == P - 1)
        written = (k < keep) | is_last[:, None, None]
        assert torch.equal(torch.isnan(hd_s), ~written)                     # nothing else was touched, nothing kept was left out
        assert torch.equal(hd_s[written], dense['harmonic_distribution'][written])
        assert float((~written).float().mean()) > 0.3                      # (the point: a good part of the tensor is never written)
        amp = amp_s.reshape(R, T)
        inh = raw_t[2].reshape(R, T).contiguous()
        for spans in (0, 1, 5):
            for split in (False, True):
                a = core.polyphonic_additive(dense['f0_hz'], amp, dense['harmonic_distribution'], None, B, N, sr, spans=spans,
                                             voice_major=vm, audible=cnt_s, split_last=split, inharm_coef=inh)
                b = core.polyphonic_additive(dense['f0_hz'], amp, hd_s, None, B, N, sr, spans=spans, voice_major=vm,
                                             audible=cnt_s, split_last=split, inharm_coef=inh)
                for x, y in zip(a if split else (a,), b if split else (b,)):
                    assert torch.equal(x, y) and bool(torch.isfinite(y).all()), (B, P, H, S, spans, split)
                # ... and the counts change nothing when the tensor is whole (they only replace exact zeros)
                c = core.polyphonic_additive(dense['f0_hz'], amp, dense['harmonic_distribution'], None, B, N, sr, spans=spans,
                                             voice_major=vm, split_last=split, inharm_coef=inh)
                for x, y in zip(a if split else (a,), c if split else (c,)):
                    assert torch.equal(x, y), (B, P, H, S, spans, split)
            a = core.polyphonic_stems(dense['f0_hz'], amp, dense['harmonic_distribution'], None, B, N, sr, spans=spans,
                                      voice_major=vm, audible=cnt_s, inharm_coef=inh)
            b = core.polyphonic_stems(dense['f0_hz'], amp, hd_s, None, B, N, sr, spans=spans, voice_major=vm, audible=cnt_s,
                                      inharm_coef=inh)
            assert torch.equal(a, b) and bool(torch.isfinite(b).all()), (B, P, H, S, spans)
    # the whole group, every dictionary form, sparse (default) against dense
    B, P, T, H, K, S, sr, L = 3, 6, 50, 128, 96, 1, 24000, 3000
    N = T * (sr // 250)
    feats = {k: torch.as_tensor(v, device='cuda') for k, v in _features(rng, B, P, T, H, K, S, L).items()}
    nz = [torch.as_tensor(rng.uniform(-1, 1, [B, N]).astype(np.float32), device='cuda') for _ in range(P)]
    gdag, _ = _build(dp, P, sr)
    pg = dp.ProcessorGroup(gdag)
    for stems in (False, 'last', True):
        set_option(monkeypatch, 'DDSPP_CONTROLS_DENSE_HD', 1)
        want = pg(feats, return_outputs_dict=True, need_stems=stems, noise=nz)
        set_option(monkeypatch, 'DDSPP_CONTROLS_DENSE_HD', None)
        got = pg(feats, return_outputs_dict=True, need_stems=stems, noise=nz)
        assert torch.equal(got['signal'], want['signal']) and bool(torch.isfinite(got['signal']).all()), stems
        if stems:
            for kk, v in want['controls']['additive']['controls'].items():
                assert torch.equal(got['controls']['additive']['controls'][kk], v), (stems, kk)
            assert torch.equal(got['controls']['additive']['signal'], want['controls']['additive']['signal'])
        if stems is True:
            assert torch.equal(got['controls']['voices']['additive'], want['controls']['voices']['additive'])
