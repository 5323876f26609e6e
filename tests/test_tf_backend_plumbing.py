"""PLUMBING ONLY -- THIS TEST PINS NOTHING.

tests/golden/tf_backend.py is the code that turns the restatement goldens into reference goldens on a host with
TensorFlow + ddsp 3.7.0 + a checkout of lrenault/ddsp-piano (DDSP_GOLDEN_BACKEND=tf python tests/golden/make_golden.py).
No such host exists for this build, so that file had never executed ("missing" #1).  Here it runs end to end
against a FAKE `tensorflow` / `ddsp` namespace and a FAKE reference checkout whose classes simply delegate to the oracle:
what is exercised is the adaptor -- module loading by path, constructor keywords, argument order, the replacement of the
unseeded tf.random.uniform by the stored noise, the numpy <-> tensor conversions, the file layout make_golden.py writes,
the `backend` tags, the per-detail report -- so that the day-one run on a real host does not die on a typo.  Because the
fakes ARE the oracle, the "tf" fixtures they produce equal the committed restatement fixtures bit for bit; that equality
says nothing about the real library.  Parity stays unpinned."""
import importlib
import os
import sys
import textwrap
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


class _T(np.ndarray):
    """A 'tensor': an ndarray with .numpy(), which is all tf_backend.py asks of one."""

    def numpy(self):
        return np.asarray(self)


def _fake_tensorflow():
    tf = types.ModuleType('tensorflow')
    tf.__version__ = '0.0-fake'
    tf.float32 = np.float32

    def convert_to_tensor(x, dtype=None):
        return np.asarray(x, dtype or np.float32).view(_T)
    tf.convert_to_tensor = convert_to_tensor
    tf.random = types.SimpleNamespace()

    def uniform(shape, minval=0, maxval=None, dtype=np.float32, seed=None, name=None):
        raise AssertionError('the unseeded draw was not replaced by the stored noise')
    tf.random.uniform = uniform
    return tf


def _fake_ddsp(O):
    ddsp = types.ModuleType('ddsp')
    ddsp.__version__ = '3.7.0-fake'
    core = types.ModuleType('ddsp.core')
    core.frequency_filter = lambda audio, magnitudes, window_size=0, padding='same': O.frequency_filter(audio, magnitudes, window_size=window_size)
    core.resample = lambda x, n, method='linear', add_endpoint=True: O.resample(x, n, method=method)
    core.angular_cumsum = lambda x, chunk_size=1000: O.angular_cumsum(x, chunk_size)
    core.exp_sigmoid = lambda x, exponent=10.0, max_value=2.0, threshold=1e-7: O.exp_sigmoid(x)
    core.fft_convolve = lambda audio, impulse_response, padding='same', delay_compensation=-1: \
        O.fft_convolve(audio, impulse_response, padding, delay_compensation)
    effects = types.ModuleType('ddsp.effects')
    effects.Reverb = O.Reverb
    processors = types.ModuleType('ddsp.processors')

    class ProcessorGroup:
        def __init__(self, dag, name='processor_group'):
            self.group = O.ProcessorGroup(dag, name)

        def __call__(self, inputs, return_outputs_dict=False):
            return self.group(inputs, return_outputs_dict=return_outputs_dict)
    processors.ProcessorGroup = ProcessorGroup
    ddsp.core, ddsp.effects, ddsp.processors = core, effects, processors
    return {'ddsp': ddsp, 'ddsp.core': core, 'ddsp.effects': effects, 'ddsp.processors': processors}


FAKE_MODULES = {
    'inharm_synth.py': """
        from oracle import ddsp_oracle as O
        MultiInharmonic, MultiAdd, exp_tanh = O.MultiInharmonic, O.MultiAdd, O.exp_tanh
    """,
    'filtered_noise_synth.py': """
        import tensorflow as tf
        from oracle import ddsp_oracle as O

        class DynamicSizeFilteredNoise(O.FilteredNoise):
            def get_signal(self, magnitudes):                       # filtered_noise_synth.py:35-42: an unseeded draw
                b, t = magnitudes.shape[0], magnitudes.shape[1]
                noise = tf.random.uniform([b, self.upsampling * t], minval=-1.0, maxval=1.0)
                return super().get_signal(magnitudes, noise=noise)
    """,
    'polyphonic_dag.py': """
        from oracle import ddsp_oracle as O
        polyphonic_dag = O.polyphonic_dag
    """,
    'surrogate_synth.py': """
        from oracle import ddsp_oracle as O
        SurrogateAdditive = O.SurrogateAdditive
    """,
    'fdn_reverb.py': """
        from oracle import ddsp_oracle as O

        class FeedbackDelayNetwork:
            def __init__(self, trainable=False, sampling_rate=16000.0, **kw):
                self.sampling_rate, self.built = sampling_rate, False

            def build(self, input_shape):
                self.built = True

            def get_ir(self, *args):
                assert self.built and len(args) == 7
                return O.fdn_get_ir(*args, sampling_rate=self.sampling_rate)
    """,
}


@pytest.fixture
def fake_host(tmp_path, monkeypatch):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import ddsp_oracle as O
    ref = tmp_path / 'fake-ddsp-piano'
    mods = ref / 'ddsp_piano' / 'modules'
    mods.mkdir(parents=True)
    for name, body in FAKE_MODULES.items():
        (mods / name).write_text(textwrap.dedent(body))
    out = tmp_path / 'golden-out'
    out.mkdir()
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k.split('.')[0] in ('tensorflow', 'ddsp', 'ddsp_piano', 'tf_backend', 'make_golden')}
    for k in saved:
        del sys.modules[k]
    sys.modules['tensorflow'] = _fake_tensorflow()
    sys.modules.update(_fake_ddsp(O))
    monkeypatch.setenv('DDSP_GOLDEN_BACKEND', 'tf')
    monkeypatch.setenv('DDSP_PIANO_REFERENCE', str(ref))
    monkeypatch.setenv('DDSP_GOLDEN_OUT', str(out))
    monkeypatch.syspath_prepend(GOLD)
    yield out
    for k in [k for k in sys.modules if k.split('.')[0] in ('tensorflow', 'ddsp', 'ddsp_piano', 'tf_backend', 'make_golden')]:
        del sys.modules[k]
    sys.modules.update({k: v for k, v in saved.items() if v is not None})


def test_tf_backend_runs_end_to_end_against_a_fake_tf_and_ddsp(fake_host, capsys):
    """(plumbing only: see the module docstring)"""
    mg = importlib.import_module('make_golden')
    mg.main()
    report = capsys.readouterr().out
    for line in ('auto_delay', 'window_crop', 'resize', 'angular_cumsum', 'angular_offsets', 'exp_sigmoid', 'initial_bias',
                 'MultiAdd order', 'bitwise: rs_linear_96'):
        assert line in report, line
    made = sorted(f for f in os.listdir(fake_host) if f.endswith('.npz'))
    assert made == sorted(['c1_mono.npz', 'c2_small.npz', 'c3_surrogate.npz', 'c4_fdn_ir.npz', 'recalled_details.npz']
                          + (['dafx22_reverb_ir.npz'] if os.path.exists(mg.REF_CKPT) else [])), made
    for f in made:
        new, old = np.load(os.path.join(fake_host, f)), np.load(os.path.join(GOLD, f))
        if f != 'dafx22_reverb_ir.npz':
            assert str(new['backend']) == 'tf' and 'fake' in str(new['backend_versions'])
        assert set(new.files) == set(old.files), (f, set(new.files) ^ set(old.files))
        for k in new.files:
            if k not in ('backend', 'backend_versions'):
                # the fakes delegate to the oracle: same bits as the committed restatement fixtures (which proves the
                # adaptor passes every argument where it belongs -- and nothing about TensorFlow)
                assert np.array_equal(new[k], old[k]), (f, k)


def test_stored_noise_replaces_the_unseeded_draw_in_call_order(fake_host):
    """(plumbing only)  _Group feeds the stored noise tensors to tf.random.uniform calls in DAG order and complains when one
    is left over."""
    from oracle import ddsp_oracle as O
    tfb = importlib.import_module('tf_backend')
    B_ = tfb.TFBackend()
    assert B_.versions == {'tensorflow': '0.0-fake', 'ddsp': '3.7.0-fake'}
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from util import synth_controls
    rng = np.random.default_rng(5)
    P, T, H, K, sr = 2, 8, 16, 9, 16000
    feats = {}
    for i in range(P):
        for k, v in synth_controls(rng, 1, T, H, S=1, K=K, silent_frac=0.0).items():
            feats[f'{k}_{i}'] = v
    noises = [rng.uniform(-1, 1, [1, T * 64]).astype(np.float32) for _ in range(P)]
    keys = dict(additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'], noise_controls=['magnitudes'],
                reverb_controls=[], n_synths=P)
    dag = B_.polyphonic_dag(B_.MultiInharmonic(name='additive', sample_rate=sr, inference=True),
                            B_.FilteredNoise(name='noise', sample_rate=sr), None, **keys)
    got = B_.ProcessorGroup(dag)(feats, extra_kwargs={'noise': [{'noise': z} for z in noises]})
    odag = O.polyphonic_dag(O.MultiInharmonic(name='additive', sample_rate=sr, inference=True),
                            O.FilteredNoise(name='noise', sample_rate=sr), None, **keys)
    want = O.ProcessorGroup(odag)(feats, extra_kwargs={'noise': [{'noise': z} for z in noises]})
    assert np.array_equal(got, want)
    swapped = B_.ProcessorGroup(dag)(feats, extra_kwargs={'noise': [{'noise': z} for z in noises[::-1]]})
    assert not np.array_equal(swapped, want)                      # order matters: voice i gets the i-th tensor
    with pytest.raises(AssertionError, match='not consumed'):
        B_.ProcessorGroup(dag)(feats, extra_kwargs={'noise': [{'noise': z} for z in noises + noises[:1]]})
