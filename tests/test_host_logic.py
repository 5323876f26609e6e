"""CPU: host side of the product (no kernels run): Processor protocol, DAG walking, DAG recognition,
table builders against the oracle, error behaviour, sharding arithmetic."""
import os

import numpy as np
import pytest
import torch

import ddsp_piano_amd as dp
from ddsp_piano_amd import core, parallel, polyphonic
from util import O


class Gain(dp.Processor):
    def __init__(self, g, name):
        super().__init__(name=name)
        self.g = g

    def get_controls(self, x):
        return {'x': x}

    def get_signal(self, x):
        return x * self.g


class Sum2(dp.Processor):
    def get_controls(self, a, b):
        return {'a': a, 'b': b}

    def get_signal(self, a, b):
        return a + b


def test_processor_call_protocol():
    p = Gain(2.0, 'g')
    x = torch.arange(6, dtype=torch.float64).reshape(2, 3)
    y = p(x)
    assert y.dtype == torch.float32 and torch.equal(y.cpu(), x.float() * 2)     # (inputs move to the GPU if there is one)
    d = p(x, return_outputs_dict=True)
    assert set(d) == {'signal', 'controls'} and set(d['controls']) == {'x'}


def test_processor_group_walks_dag_with_nested_keys():
    a, b, s = Gain(2.0, 'a'), Gain(3.0, 'b'), Sum2('sum')
    pg = dp.ProcessorGroup([(a, ['in']), (b, ['a/signal']), (a, ['b/signal']), (s, ['a/signal', 'b/controls/x'])])
    x = torch.ones(1, 4)
    out = pg({'in': x}, return_outputs_dict=True)
    assert torch.equal(out['signal'].cpu(), x * 12 + x * 2)        # a re-used: outputs['a'] overwritten
    ctl = out['controls']
    assert ctl['out'] is ctl['sum'] and torch.equal(ctl['in'].cpu(), x) and 'inputs' in ctl
    assert [p.name for p in pg.processors] == ['a', 'b', 'sum'] and pg.a is a
    assert torch.equal(pg({'in': x}), out['signal'])
    assert torch.equal(pg.get_signal(pg.get_controls({'in': x})), out['signal'])
    with pytest.raises(KeyError):
        pg({'nope': x})
    with pytest.raises(TypeError):
        dp.ProcessorGroup([('not a processor', ['in'])])


def _dag(P, with_reverb=True, reverb=None):
    add = dp.MultiInharmonic(name='additive', sample_rate=24000)
    nz = dp.DynamicSizeFilteredNoise(name='noise', sample_rate=24000)
    rv = reverb if reverb is not None else (dp.Reverb() if with_reverb else None)
    return dp.polyphonic_dag(add, nz, rv, additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'],
                             noise_controls=['magnitudes'], reverb_controls=['reverb_ir'] if rv else [], n_synths=P)


def test_polyphonic_dag_matches_reference_node_list():
    dag = _dag(3)
    odag = O.polyphonic_dag(O.MultiInharmonic(name='additive'), O.FilteredNoise(name='noise'), O.Reverb(),
                            additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'],
                            noise_controls=['magnitudes'], reverb_controls=['reverb_ir'], n_synths=3)
    assert [(n[0].name, list(n[1])) for n in dag] == [(n[0].name, list(n[1])) for n in odag]


def test_fast_path_recognition():
    plan = polyphonic.recognise(_dag(4))
    assert plan is not None and plan.n_synths == 4 and plan.reverb_keys == ['reverb_ir']
    assert polyphonic.recognise(_dag(2, with_reverb=False)).reverb is None
    assert polyphonic.recognise(_dag(2, reverb=dp.FeedbackDelayNetworkApply())) is not None
    dag = _dag(3)
    dag[5] = (dag[5][0], ['noise/signal', 'additive/signal'])              # not the add chain
    assert polyphonic.recognise(dag) is None
    assert polyphonic.recognise([(Gain(1.0, 'g'), ['x'])] * 3) is None
    a, b = Gain(1.0, 'a'), Gain(1.0, 'b')
    assert dp.ProcessorGroup([(a, ['x']), (b, ['a/signal'])])({'x': torch.ones(2)}).shape == (2,)


def _default_model_dag(P, with_reverb=True):
    """ddsp_piano/default_model.py:44-80."""
    nz = dp.DynamicSizeFilteredNoise(name='noise', sample_rate=16000)
    add = dp.MultiInharmonic(name='additive', sample_rate=16000, inference=True)
    ctl = ['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz']
    dag = [(nz, ['magnitudes_0']), (add, [c + '_0' for c in ctl]), (dp.Add(name='add_0'), ['noise/signal', 'additive/signal'])]
    for i in range(1, P):
        dag += [(add, [c + f'_{i}' for c in ctl]), (nz, [f'magnitudes_{i}']),
                (dp.Add(name=f'sub_add_{i}'), ['noise/signal', 'additive/signal']),
                (dp.Add(name=f'add_{i}'), [f'add_{i - 1}/signal', f'sub_add_{i}/signal'])]
    if with_reverb:
        dag.append((dp.Reverb(trainable=False, reverb_length=100), [f'add_{P - 1}/signal', 'reverb_ir']))
    return dag


def test_default_model_node_list_is_recognised():
    """Round 4: the node list default_model.py builds (noise first, explicit Add nodes) takes the batched route too."""
    for P in (1, 2, 5):
        for rv in (True, False):
            plan = polyphonic.recognise(_default_model_dag(P, rv))
            assert plan is not None and plan.shape == 'default_model' and plan.n_synths == P
            assert (plan.reverb is not None) == rv and plan.reverb_keys == (['reverb_ir'] if rv else [])
            assert plan.noise_keys == [f'magnitudes_{i}' for i in range(P)] and plan.additive_keys[P - 1][3] == f'f0_hz_{P - 1}'
            assert [a.name for a in plan.adds] == [f'add_{i}' for i in range(P)] and plan.add is plan.adds[-1]
    assert polyphonic.recognise(_dag(3)).shape == 'gin'
    bad = _default_model_dag(3)
    bad[6] = (bad[6][0], ['add_0/signal', 'sub_add_2/signal'])         # add_1 fed from the wrong pair
    assert polyphonic.recognise(bad) is None
    bad = _default_model_dag(3)
    bad[-1] = (bad[-1][0], ['add_1/signal', 'reverb_ir'])                # reverb not on the last add
    assert polyphonic.recognise(bad) is None
    bad = _default_model_dag(3)
    bad[9] = (bad[5][0], bad[9][1])                                       # an Add object used twice
    assert polyphonic.recognise(bad) is None
    with pytest.raises(ValueError):
        dp.NativeGroup(dp.ProcessorGroup(_default_model_dag(2)), {})     # the one-call driver keeps to polyphonic_dag's shape


def test_stack_voices_zero_copy_on_cpu_falls_back_to_copy():
    base = torch.randn(2, 3, 5, 4)
    views = [base[:, i] for i in range(3)]
    st, vm = polyphonic._stack_voices(views)            # CPU tensors: copied, voice major by default
    assert vm is True and st.shape == (6, 5, 4) and torch.equal(st.cpu(), base.transpose(0, 1).reshape(6, 5, 4))
    st, vm = polyphonic._stack_voices(views, False)
    assert vm is False and torch.equal(st.cpu(), base.reshape(6, 5, 4))


def test_tables_match_the_oracle():
    for T, U in [(750, 96), (50, 64), (40, 32), (20, 128), (33, 100)]:
        lo, hi, w, aligned = core._linear_tables_np(T, T * U)
        olo, ohi, ow = O.linear_resample_positions(T, T * U)
        assert np.array_equal(lo, olo) and np.array_equal(hi, ohi) and np.array_equal(w, ow) and aligned
    lo, hi, w, aligned = core._linear_tables_np(37, 1000)
    assert not aligned and np.array_equal(w, O.linear_resample_positions(37, 1000)[2])
    for n in (64, 128, 192, 257, 190, 7):
        assert np.array_equal(core._hann_window_np(n), O.hann_window(n))
    assert core.fused_synthesis_supported(750, 72000) and not core.fused_synthesis_supported(37, 1000)
    assert not core.fused_synthesis_supported(20, 2000)            # U = 100 is not a multiple of 8


@pytest.mark.parametrize('K,ws', [(96, 257), (64, 257), (32, 257), (128, 257), (200, 257), (65, 0), (129, 257)])
def test_fir_matrix_is_frequency_impulse_response(K, ws):
    rng = np.random.default_rng(K)
    mags = rng.uniform(0, 2, [5, K]).astype(np.float32)
    m = core._fir_matrix_np(K, ws)
    ref = O.frequency_impulse_response(mags, ws)
    assert m.shape == (K, ref.shape[-1])
    np.testing.assert_allclose(mags.astype(np.float64) @ m.astype(np.float64), ref, atol=3e-7)
    uniq, mirror = core._fir_symmetry_np(K, ws)
    covered = set(uniq.tolist()) | set(int(x) for x in mirror if x >= 0)
    assert covered == set(range(m.shape[1]))
    for u, mi in zip(uniq, mirror):
        if mi >= 0:
            assert np.abs(m[:, u] - m[:, mi]).max() < 1e-6 * np.abs(m).max()


def test_scale_fn_recognition():
    import functools
    assert core.scale_kind(None)[0] == 0
    assert core.scale_kind(dp.exp_sigmoid)[0] == 1 and core.scale_kind(dp.exp_tanh)[0] == 2
    k = core.scale_kind(functools.partial(dp.exp_sigmoid, max_value=3.0))
    assert k[0] == 1 and k[1]['max_value'] == 3.0
    assert core.scale_kind(lambda x: x) is None
    assert core.scale_kind(functools.partial(dp.exp_sigmoid, bogus=1)) is None


def test_shape_errors_are_raised_before_any_kernel():
    x = torch.zeros(2, 10, 4)
    with pytest.raises(ValueError):
        core.upsample_with_windows(torch.zeros(10, 4), 100)
    with pytest.raises(ValueError):
        core.resample(x, 95, method='window')
    with pytest.raises(ValueError):
        core.resample(x, 100, method='bogus')
    with pytest.raises(ValueError):
        core.fft_convolve(torch.zeros(2, 100), torch.zeros(3, 10))
    with pytest.raises(ValueError):
        core.fft_convolve(torch.zeros(2, 100), torch.zeros(2, 60, 10))
    with pytest.raises(ValueError):
        core.fft_convolve(torch.zeros(2, 100), torch.zeros(2, 10), padding='full')
    with pytest.raises(ValueError):
        dp.Reverb().get_controls(torch.zeros(2, 100))
    with pytest.raises(ValueError):
        dp.InHarmonic().get_controls(torch.zeros(2, 5, 1), torch.zeros(2, 5, 8), torch.zeros(2, 5, 1),
                                     torch.zeros(2, 5, 2))
    with pytest.raises(ValueError):
        dp.MultiInharmonic().get_controls(torch.zeros(2, 5, 2), torch.zeros(2, 5, 8), torch.zeros(2, 5, 1),
                                          torch.zeros(2, 5, 2))
    with pytest.raises(ValueError):
        core.cos_oscillator_bank(torch.zeros(1, 8, 4), torch.zeros(1, 8, 5))


def test_no_cpu_fallback():
    """The product refuses CPU buffers instead of quietly computing somewhere else."""
    if torch.cuda.is_available():
        pytest.skip('GPU box: tensors are moved to the device')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        core.resample(torch.zeros(1, 10, 4), 640)
    with pytest.raises(RuntimeError):
        dp.MultiInharmonic(sample_rate=16000).get_signal(torch.zeros(1, 10, 1), torch.zeros(1, 10, 8),
                                                         torch.zeros(1, 10, 8), torch.zeros(1, 10, 1))


def test_shard_ranges():
    for B, W in [(512, 8), (64, 1), (10, 4), (3, 8), (0, 2)]:
        rs = [parallel.shard_range(B, W, r) for r in range(W)]
        assert rs[0][0] == 0 and rs[-1][1] == B
        assert all(rs[i][1] == rs[i + 1][0] for i in range(W - 1))
        sizes = [b - a for a, b in rs]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_range(8, 2, 2)
    feats = {'a_0': torch.zeros(8, 5, 1), 'reverb_ir': torch.zeros(8, 100), 'flag': 3}
    sh = parallel.shard_features(feats, 4, 1)
    assert sh['a_0'].shape == (2, 5, 1) and sh['reverb_ir'].shape == (2, 100) and sh['flag'] == 3
    with pytest.raises(ValueError):
        parallel.shard_features({'a': torch.zeros(8, 5, 1), 'b': torch.zeros(6, 5, 1)}, 2, 0)
    # an impulse response shared by every row ([L] or [1, L], both accepted by ddsp.effects.Reverb) is not sharded
    shared = {'a_0': torch.zeros(8, 5, 1), 'ir1': torch.zeros(100), 'ir2': torch.zeros(1, 100), 'ir3': torch.zeros(8, 100)}
    sh = parallel.shard_features(shared, 4, 3)
    assert sh['ir1'].shape == (100,) and sh['ir2'].shape == (1, 100) and sh['ir3'].shape == (2, 100) and sh['a_0'].shape[0] == 2


def test_parallelizer_merges_and_unmerges_like_the_reference():
    """sub_modules.py:527-602: globals are repeated / transposed into [P * B, ...]; mono keys come back as
    per-voice VIEWS of the merged buffer."""
    P, B, T = 3, 2, 5
    par = dp.Parallelizer(n_synths=P)
    feats = {'conditioning': torch.arange(B * T * P * 2, dtype=torch.float32).reshape(B, T, P, 2),
             'context': torch.randn(B, T, 7), 'global_inharm': torch.randn(B, T), 'global_detuning': torch.randn(B, T)}
    cond0, ctx0 = feats['conditioning'].clone(), feats['context'].clone()
    out = par(dict(feats), parallelize=True)
    assert par.batch_size == B
    assert out['conditioning'].shape == (P * B, T, 2) and out['context'].shape == (P * B, T, 7)
    assert out['global_inharm'].shape == (P * B, T)
    for i in range(P):                                  # voice-major rows: [i * B, (i + 1) * B) is voice i
        assert torch.equal(out['conditioning'][i * B:(i + 1) * B], cond0[:, :, i])
        assert torch.equal(out['context'][i * B:(i + 1) * B], ctx0)
    merged = {k: torch.randn(P * B, T, c) for k, c in (('f0_hz', 1), ('inharm_coef', 1), ('amplitudes', 1),
                                                       ('harmonic_distribution', 8), ('magnitudes', 4))}
    keep = {k: v for k, v in merged.items()}
    un = par(dict(merged), parallelize=False)
    for k, v in keep.items():
        assert un[k].shape == (P, B) + tuple(v.shape[1:])
        for i in range(P):
            assert torch.equal(un[f'{k}_{i}'], v[i * B:(i + 1) * B])
            assert un[f'{k}_{i}'].data_ptr() == v[i * B:(i + 1) * B].data_ptr()       # a view, not a copy


def test_time_shard_ranges_are_whole_blocks():
    """One long file over several GPUs: every rank's frame range starts on a 1000-sample chunk boundary (a multiple of
    125 frames) and the ranges tile the file."""
    from ddsp_piano_amd import parallel, streaming
    assert [streaming.block_frames(u) for u in (32, 64, 96, 128, 192)] == [125] * 5
    for t, world in ((34000, 8), (1130, 2), (1000, 3), (100, 4)):
        rs = [parallel.time_shard_range(t, world, r) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == t
        for (lo, hi), (lo2, _) in zip(rs, rs[1:]):
            assert hi == lo2 and lo % 125 == 0 and hi % 125 == 0
    with pytest.raises(ValueError):
        parallel.time_shard_range(1000, 2, 2)


def test_bench_launcher_logic(monkeypatch):
    """`python bench.py --gpus N` outside a launcher starts N ranks itself, refuses when the box has fewer GPUs, and a
    launcher that started a different number of ranks is an error -- never a silent one-GPU number."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    assert bench.launcher_command(1, {}, [], 0) is None
    assert bench.launcher_command(2, {'WORLD_SIZE': '2'}, [], 0) is None
    with pytest.raises(SystemExit, match='1 GPU'):
        bench.launcher_command(2, {}, ['--gpus', '2'], 1)
    with pytest.raises(SystemExit, match='WORLD_SIZE=2'):
        bench.launcher_command(4, {'WORLD_SIZE': '2'}, [], 8)
    cmd = bench.launcher_command(8, {}, ['--gpus', '8', '--steps', '3'], 8)
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=8' in cmd
    assert cmd[-4:] == ['--gpus', '8', '--steps', '3'] and '127.0.0.1' in cmd
    args = bench.parse(['--gpus', '2', '--no-single-stream'])
    assert args.no_extras and args.call_form == 'outputs_dict'
    # the dry run of the multi-rank flow on a box with fewer GPUs is explicit (and marked in the JSON line)
    cmd = bench.launcher_command(2, {'DDSPP_BENCH_SHARE_GPU': '1'}, ['--gpus', '2'], 1)
    assert cmd is not None and '--nproc-per-node=2' in cmd


def test_bench_extras_levels_and_gpu_sampler_without_a_gpu():
    """Round 5: `--extras default | full | none` (and the old `--no-extras`), `--sustain-seconds`; the clock / power sampler of the
    `sustained` measurement must never take the timing down with it: on a box without readable amdgpu hwmon files (this one) it
    falls back to rocm-smi, reads nothing, and reports an empty summary."""
    import importlib
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    a = bench.parse([])
    assert a.extras == 'default' and not a.no_extras and a.sustain_seconds == 5.0 and a.gpus == 1
    assert bench.parse(['--extras', 'full']).extras == 'full' and bench.parse(['--no-extras']).no_extras
    with pytest.raises(SystemExit):
        bench.parse(['--extras', 'everything'])
    with bench.GpuSampler(0, period=0.01) as smp:
        time.sleep(0.05)
    out = smp.summary()
    assert out['source'] in ('rocm-smi',) or out['source'].startswith('sysfs hwmon')
    assert out['n'] == len(smp.samples) and ('sclk_mhz' in out) == any(r[1] == r[1] for r in smp.samples)
    assert smp.summary(t_from=time.perf_counter() + 1.0)['n'] == 0
    assert bench.ms_summary([1.0, 3.0, 2.0]) == {'median': 2.0, 'min': 1.0, 'max': 3.0, 'n': 3}


def test_plan_cache_is_bounded_and_respects_pins(monkeypatch):
    """rocFFT plans are library-owned handles: the cache destroys the least recently used ones when it is full, never a
    plan pinned by an unfinished two-phase convolution, and clear() (atexit) tolerates a library that is gone."""
    destroyed = []

    class FakeLib:
        def fake_destroy(self, h):
            destroyed.append(h)
    monkeypatch.setattr(core, '_lib_', lambda: FakeLib())
    cache = core._PlanCache('fake_destroy', maxsize=2)
    a = cache.get('a', lambda: 'A')
    cache.pin(a, +1)
    cache.get('b', lambda: 'B')
    cache.get('c', lambda: 'C')                     # full: 'a' is pinned, so 'b' (the oldest unpinned) goes
    assert destroyed == ['B'] and len(cache) == 2
    assert cache.get('a', lambda: 'A2')[0] == 'A'   # still the same plan
    cache.pin(a, -1)
    cache.get('d', lambda: 'D')
    assert 'C' in destroyed and len(cache) == 2
    cache.clear()
    assert len(cache) == 0 and set(destroyed) == {'A', 'B', 'C', 'D'}
    monkeypatch.setattr(core, '_lib_', lambda: (_ for _ in ()).throw(RuntimeError('gone')))
    cache.clear()                                   # no exception at interpreter exit


def test_plan_cache_pins_what_a_capture_recorded():
    """a captured HIP graph replays rocFFT executions without calling the plan cache again, so the plans it
    used must not be LRU-evicted; get(pin=True) closes the window between look-up and use."""
    from ddsp_piano_amd import core
    cache = core._PlanCache('ddspp_irfft_plan_destroy', maxsize=2)      # handles are NULL: destroy is a no-op
    created = []

    def make(tag):
        def create():
            created.append(tag)
            return None
        return create
    with cache.record() as used:
        e1 = cache.get('k1', make('k1'))
    assert used == [e1] and e1[2] == 1                                  # pinned once by the recorder
    for k in ('k2', 'k3', 'k4'):
        cache.get(k, make(k))
    assert set(cache._entries) == {'k1', 'k4'}                          # k1 stays although it is the oldest
    cache.unpin_all(used)
    cache.get('k5', make('k5'))
    assert set(cache._entries) == {'k4', 'k5'}
    e = cache.get('k6', make('k6'), pin=True)
    assert e[2] == 1
    cache.get('k7', make('k7'))
    cache.get('k8', make('k8'))
    assert set(cache._entries) == {'k6', 'k8'}                          # in use: not evictable
    cache.pin(e, -1)
    cache.get('k9', make('k9'))
    assert set(cache._entries) == {'k8', 'k9'}
    assert created == ['k1', 'k2', 'k3', 'k4', 'k5', 'k6', 'k7', 'k8', 'k9']


def test_linear_exact_frames_bound():
    """past this many frames float32(n) * float32(1 / U) rounds across a frame boundary (round 4: the samples
    concerned are marked in the weight table, test_walk_weights_...; streaming no longer refuses)."""
    import numpy as np
    from ddsp_piano_amd import core
    f = core.linear_exact_frames(96)
    assert 2 ** 17 <= f <= 2 ** 17 + 8
    scale = np.float32(1.0) / np.float32(96)
    k = np.arange(1, f, dtype=np.int64)
    assert (np.floor((k * 96).astype(np.float32) * scale) == k).all()
    assert (np.floor((k * 96 - 1).astype(np.float32) * scale) == k - 1).all()
    bad_first = np.floor(np.float32(f * 96) * scale) != f
    bad_last = np.floor(np.float32(f * 96 - 1) * scale) != f - 1
    assert bad_first or bad_last
    assert core.linear_exact_frames(32) > core.linear_exact_frames(96) >= core.linear_exact_frames(192)


def test_walk_weights_mark_the_samples_that_take_the_next_row():
    """Round 4: past linear_exact_frames the reference's resize takes rows (t + 1, t + 1) for the last sample(s) of a frame;
    the frame-walking kernels get those samples marked (core._walk_weights_np == ddspp_walk_weights_host) and substitute
    x[t + 1] exactly.  The mark reproduces the three-operator tables sample for sample."""
    import ctypes
    import numpy as np
    from ddsp_piano_amd import _lib, core
    lib = _lib.load()
    U = 96
    f = core.linear_exact_frames(U)
    # (1) a short signal: no marks, the plain fractional parts
    w, ok = core._walk_weights_np(750, 72000)
    assert ok and np.array_equal(w, core._linear_tables_np(750, 72000)[2]) and not (w == 1).any()
    # (2) a 150 000-frame file: marked samples exist, all of them last samples of a frame past `f`, and the walk
    # reproduces the table: x[lo] + (x[hi] - x[lo]) w == (mark ? x[t + 1] : x[t] + (x[t + 1] - x[t]) w)
    T = 150000
    N = T * U
    lo, hi, wt, aligned = core._linear_tables_np(T, N)
    w, ok = core._walk_weights_np(T, N)
    assert ok and not aligned and core.fused_synthesis_supported(T, N)
    marked = np.nonzero(w == 1)[0]
    assert marked.size > 1000 and marked.min() // U >= f - 1 and (marked % U >= U - 8).all()
    n = np.arange(N)
    t = n // U
    assert np.array_equal(lo[marked], np.minimum(t[marked] + 1, T - 1)) and (wt[marked] == 0).all()
    rest = np.ones(N, bool)
    rest[marked] = False
    assert np.array_equal(lo[rest], t[rest]) and np.array_equal(w[rest], wt[rest])
    rng = np.random.default_rng(5)
    x = rng.uniform(20, 4000, T).astype(np.float32)
    ref = (x[lo] + (x[hi] - x[lo]) * wt).astype(np.float32)
    t1 = np.minimum(t + 1, T - 1)
    walk = np.where(w == 1, x[t1], (x[t] + (x[t1] - x[t]) * w).astype(np.float32))
    assert np.array_equal(ref, walk)
    # (3) the C builder gives the same table, also for a piece at an absolute position
    for first, cnt in ((0, N), (140000 * U, 2000 * U)):
        wc = np.empty(cnt, np.float32)
        flag = ctypes.c_int(-1)
        tc = T if first == 0 else cnt // U
        assert lib.ddspp_walk_weights_host(tc, tc * U, 0, first, cnt, wc.ctypes.data_as(ctypes.c_void_p), ctypes.byref(flag)) == 0
        wp, okp = core._walk_weights_np(tc, tc * U, 'legacy', first, cnt)
        assert flag.value == 1 and okp and np.array_equal(wc, wp)
        assert np.array_equal(core.walk_weights(tc, tc * U, 'cpu', first).numpy(), wp)
    assert np.array_equal(wp, w[140000 * U:142000 * U])                 # the piece's table is the file's
    # (4) not walkable: a ratio that is no whole number, the half-pixel rule
    assert not core._walk_weights_np(37, 1000)[1] and not core._walk_weights_np(750, 72000, 'half_pixel')[1]
    flag = ctypes.c_int(-1)
    wc = np.empty(1000, np.float32)
    assert lib.ddspp_walk_weights_host(37, 1000, 0, 0, 1000, wc.ctypes.data_as(ctypes.c_void_p), ctypes.byref(flag)) == 0
    assert flag.value == 0


def test_per_voice_tensors_are_recognised_as_slices_of_one_buffer():
    """polyphonic._same_buffer_slices (the zero-copy test of the per-voice keys the Parallelizer hands over): slices at
    exact multiples inside one storage; the storage is asked of the first and the last tensor only."""
    import torch
    from ddsp_piano_amd import polyphonic
    whole = torch.arange(4 * 3 * 5 * 2, dtype=torch.float32).reshape(4, 3, 5, 2)          # [P, B, T, C]
    voices = [whole[i] for i in range(4)]
    assert polyphonic._same_buffer_slices(voices, 3 * 5 * 2)
    assert not polyphonic._same_buffer_slices(voices, 5 * 2)                               # another step
    assert not polyphonic._same_buffer_slices([voices[0], voices[2], voices[1], voices[3]], 3 * 5 * 2)   # another order
    assert not polyphonic._same_buffer_slices(voices[:3] + [voices[3].clone()], 3 * 5 * 2)               # another storage
    seg = torch.arange(3 * 4 * 5 * 2, dtype=torch.float32).reshape(3, 4, 5, 2)            # [B, P, T, C]
    assert polyphonic._same_buffer_slices([seg[:, i] for i in range(4)], 5 * 2)
    # host tensors (or mixed shapes) are stacked by copy, voice major by default
    rows, vm = polyphonic._stack_voices(voices)
    assert vm is True and rows.shape == (12, 5, 2) and torch.equal(rows, whole.reshape(12, 5, 2))
    rows, vm = polyphonic._stack_voices([seg[:, i] for i in range(4)], False)
    assert vm is False and torch.equal(rows, seg.reshape(12, 5, 2))


def test_voice_sums_follow_the_batch_size():
    """polyphonic.pick_voice_sums (mirrored in csrc/group.cpp): sums of eight voices at batch 64, per-voice rows for a single
    3 s segment (50 units of eight voices would leave the chip idle), the forced form of the tests."""
    from ddsp_piano_amd.polyphonic import pick_voice_sums
    assert pick_voice_sums(64, 16, 750) == 8 and pick_voice_sums(1, 16, 750) == 1
    assert pick_voice_sums(8, 16, 750) == 4            # 8 x 4 x 25 = 800 units with sums of four, 400 with eight
    assert pick_voice_sums(1, 16, 34000) == 8          # a 136 s file: 2 x 1134 units
    assert pick_voice_sums(64, 6, 750) == 2 and pick_voice_sums(64, 5, 750) == 1
    assert pick_voice_sums(1, 16, 10, forced=8) == 8 and pick_voice_sums(1, 6, 10, forced=8) == 2
    assert pick_voice_sums(1, 16, 10, forced=4) == 4
