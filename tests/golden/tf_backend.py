"""DDSP_GOLDEN_BACKEND=tf: the golden generator's view of the REAL reference (TensorFlow + ddsp 3.7.0 + the
lrenault/ddsp-piano checkout), with the same small interface make_golden.py uses on oracle/ddsp_oracle.py.

Nothing here can run in the build container or on the GPU box (no TensorFlow, no ddsp, no network); it is the
"one command" that turns the restatement goldens into reference goldens on any host that has

    pip install ddsp==3.7.0          # pulls tensorflow, gin-config, ...   (reference README.md:11-15)
    git clone https://github.com/lrenault/ddsp-piano  (or DDSP_PIANO_REFERENCE=/path/to/checkout)

    DDSP_GOLDEN_BACKEND=tf python tests/golden/make_golden.py

It loads the reference's own modules -- ddsp_piano/modules/inharm_synth.py (MultiInharmonic, MultiAdd),
filtered_noise_synth.py:12-42 (DynamicSizeFilteredNoise), polyphonic_dag.py -- next to ddsp.effects.Reverb and
ddsp.processors.ProcessorGroup, feeds them the same seeded inputs, and replaces the unseeded
``tf.random.uniform`` draw of the noise synthesiser (filtered_noise_synth.py:39-40) by the stored noise tensors
(same call order as the DAG: voice 0, 1, ...).  Only the package __init__ files are bypassed (they import the
training / data-pipeline modules, which need note_seq etc.); every line that computes audio is the reference's.
"""
import importlib
import os
import sys
import types

import numpy as np


def _to_np(x):
    if isinstance(x, dict):
        return {k: _to_np(v) for k, v in x.items()}
    if hasattr(x, 'numpy'):
        return np.asarray(x.numpy())
    return x


class _Proc:
    """numpy-in / numpy-out view of one reference processor (the TF object stays reachable as .tf)."""

    def __init__(self, tf_processor, tf):
        self.tf = tf_processor
        self._tf = tf
        self.name = tf_processor.name

    def _t(self, x):
        return self._tf.convert_to_tensor(np.asarray(x, np.float32))

    def get_controls(self, *args):
        return _to_np(self.tf.get_controls(*[self._t(a) for a in args]))

    def get_signal(self, **controls):
        return _to_np(self.tf.get_signal(**{k: self._t(v) for k, v in controls.items()}))

    def __call__(self, *args):
        return _to_np(self.tf(*[self._t(a) for a in args]))


class _Group:
    def __init__(self, dag, tf, ddsp):
        self._tf = tf
        self.group = ddsp.processors.ProcessorGroup(dag=dag)

    def __call__(self, feats, return_outputs_dict=False, extra_kwargs=None):
        tf = self._tf
        queue = []
        for calls in (extra_kwargs or {}).values():            # {noise processor name: [{'noise': z}, ...]}
            queue.extend(np.asarray(kw['noise'], np.float32) for kw in calls)
        original = tf.random.uniform

        def stored_noise(shape, minval=0, maxval=None, dtype=tf.float32, seed=None, name=None):
            z = queue.pop(0)
            assert tuple(int(s) for s in shape) == tuple(z.shape), (shape, z.shape)
            assert float(minval) == -1.0 and float(maxval) == 1.0
            return tf.convert_to_tensor(z, dtype)

        tf.random.uniform = stored_noise
        try:
            out = self.group({k: tf.convert_to_tensor(np.asarray(v, np.float32)) for k, v in feats.items()},
                             return_outputs_dict=True)
        finally:
            tf.random.uniform = original
        assert not queue, f'{len(queue)} stored noise tensor(s) were not consumed'
        out = _to_np(out)
        return out if return_outputs_dict else out['signal']


class TFBackend:
    name = 'tf'

    def __init__(self, reference_root=None):
        import tensorflow as tf
        import ddsp
        import ddsp.effects
        import ddsp.processors
        root = reference_root or os.environ.get('DDSP_PIANO_REFERENCE', '/root/reference')
        if not os.path.isdir(os.path.join(root, 'ddsp_piano', 'modules')):
            raise SystemExit(f'DDSP_PIANO_REFERENCE={root} is not a checkout of lrenault/ddsp-piano')
        for name, sub in (('ddsp_piano', 'ddsp_piano'), ('ddsp_piano.modules', os.path.join('ddsp_piano', 'modules'))):
            if name not in sys.modules:            # namespace stand-ins: sub-modules import, the __init__ files do not run
                pkg = types.ModuleType(name)
                pkg.__path__ = [os.path.join(root, sub)]
                sys.modules[name] = pkg
        self.tf, self.ddsp = tf, ddsp
        self.inharm = importlib.import_module('ddsp_piano.modules.inharm_synth')
        self.noise = importlib.import_module('ddsp_piano.modules.filtered_noise_synth')
        self.dag = importlib.import_module('ddsp_piano.modules.polyphonic_dag')
        self.surrogate = importlib.import_module('ddsp_piano.modules.surrogate_synth')
        self.versions = {'tensorflow': tf.__version__, 'ddsp': getattr(ddsp, '__version__', 'unknown')}

    # the constructors make_golden.py calls, keyword for keyword those of the gin files (maestro-v2.gin:155-164)
    def MultiInharmonic(self, **kw):
        return _Proc(self.inharm.MultiInharmonic(**kw), self.tf)

    def FilteredNoise(self, **kw):
        return _Proc(self.noise.DynamicSizeFilteredNoise(**kw), self.tf)

    def SurrogateAdditive(self, **kw):                 # configs/surrogate.gin:121-127
        if getattr(kw.get('scale_fn'), '__func__', None) is TFBackend.exp_tanh:      # make_golden passes B_.exp_tanh: the reference's own
            kw['scale_fn'] = self.inharm.exp_tanh
        return _Proc(self.surrogate.SurrogateAdditive(**kw), self.tf)

    def Reverb(self, **kw):
        kw.setdefault('trainable', False)
        return _Proc(self.ddsp.effects.Reverb(**kw), self.tf)

    def polyphonic_dag(self, additive, noise, reverb=None, **kw):
        return self.dag.polyphonic_dag(additive.tf, noise.tf, None if reverb is None else reverb.tf,
                                       **{k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in kw.items()})

    def ProcessorGroup(self, dag):
        return _Group(dag, self.tf, self.ddsp)

    def fdn_get_ir(self, input_gain, output_gain, gain_allpass, delays_allpass, time_rev_0_sec, alpha_tone, early_ir,
                   sampling_rate=16000.0):
        """FeedbackDelayNetwork(trainable=False).get_ir -- fdn_reverb.py:339-360 with the layer's own delay values and mixing
        matrix (build(), :92-125): one network, the parameters as arrays (input / output gains [D], all-pass gains and delays
        [D, 4], T60 and tone scalars, early reflections [E])."""
        fdn = importlib.import_module('ddsp_piano.modules.fdn_reverb')
        layer = fdn.FeedbackDelayNetwork(trainable=False, sampling_rate=sampling_rate)
        layer.build(None)
        t = lambda x: self.tf.convert_to_tensor(np.asarray(x, np.float32))       # noqa: E731
        return _to_np(layer.get_ir(t(input_gain), t(output_gain), t(gain_allpass), t(delays_allpass), t(time_rev_0_sec),
                                   t(alpha_tone), t(early_ir)))

    # single operators, for the per-detail report (which recollection does the real library match?)
    def frequency_filter(self, audio, magnitudes, window_size):
        t = self.tf.convert_to_tensor
        return _to_np(self.ddsp.core.frequency_filter(t(audio), t(magnitudes), window_size=window_size))

    def resample(self, x, n, method='linear'):
        return _to_np(self.ddsp.core.resample(self.tf.convert_to_tensor(x), n, method=method))

    def angular_cumsum(self, x):
        return _to_np(self.ddsp.core.angular_cumsum(self.tf.convert_to_tensor(x)))

    def exp_sigmoid(self, x):
        return _to_np(self.ddsp.core.exp_sigmoid(self.tf.convert_to_tensor(x)))

    def fft_convolve(self, audio, impulse_response, padding='same', delay_compensation=-1):
        t = self.tf.convert_to_tensor
        return _to_np(self.ddsp.core.fft_convolve(t(audio), t(impulse_response), padding=padding,
                                                  delay_compensation=delay_compensation))

    def exp_tanh(self, x):
        return _to_np(self.inharm.exp_tanh(self.tf.convert_to_tensor(x)))

    def multi_add(self, signals):
        """The reference's MultiAdd processor on the given signals (inharm_synth.py:296-309)."""
        t = self.tf.convert_to_tensor
        return _to_np(self.inharm.MultiAdd()(*[t(np.asarray(s, np.float32)) for s in signals]))
