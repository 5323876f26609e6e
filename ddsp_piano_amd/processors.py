"""The ddsp Processor protocol (ddsp.processors.Processor / ProcessorGroup / Add, ddsp.dags.DAGLayer)
re-stated without Keras: same constructor arguments, same ``get_controls`` -> ``get_signal`` call
flow, same output dictionaries, so that it sits where the reference's ProcessorGroup sits
(ddsp_piano/modules/piano_model.py:160-164; ddsp_piano/default_model.py:20-85;
ddsp_piano/configs/maestro-v2.gin:144-153).
"""
from __future__ import annotations

import torch

from . import core


class Processor:
    """ddsp.processors.Processor: ``__call__ = get_signal(**get_controls(*args, **kwargs))``."""

    def __init__(self, name, trainable=False):
        self.name = name
        self.trainable = trainable

    def __call__(self, *args, return_outputs_dict=False, **kwargs):
        # "Convert input tensors to float32" (ddsp.processors.Processor.call)
        args = [core.tf_float32(a) if _is_tensor_like(a) else a for a in args]
        kwargs = {k: core.tf_float32(v) if _is_tensor_like(v) else v for k, v in kwargs.items()}
        controls = self.get_controls(*args, **kwargs)
        signal = self.get_signal(**controls)
        if return_outputs_dict:
            return dict(signal=signal, controls=controls)
        return signal

    def get_controls(self, *args, **kwargs):
        raise NotImplementedError

    def get_signal(self, *args, **kwargs):
        raise NotImplementedError


def _is_tensor_like(x):
    if isinstance(x, torch.Tensor):
        return True
    try:
        import numpy as np
        return isinstance(x, np.ndarray)
    except Exception:  # noqa: BLE001
        return False


class Add(Processor):
    """ddsp.processors.Add (default_model.py:56-74)."""

    def __init__(self, name='add'):
        super().__init__(name=name)

    def get_controls(self, signal_one, signal_two):
        return {'signal_one': signal_one, 'signal_two': signal_two}

    def get_signal(self, signal_one, signal_two):
        return core.add_signals([signal_one, signal_two])


def _nested_lookup(key, dictionary):
    """ddsp.core.nested_lookup: 'a/b/c' -> dictionary['a']['b']['c']."""
    out = dictionary
    for k in key.split('/'):
        out = out[k]
    return out


class ProcessorGroup:
    """ddsp.processors.ProcessorGroup(dag, name): string-keyed DAG of processors.

    ``dag`` is a list of ``(processor, [input_key, ...])``; keys are looked up, '/'-nested, in the
    running outputs dict that starts as the input features.  ``outputs[processor.name]`` is
    overwritten each time a processor is reused (polyphonic_dag.py re-uses three objects for all
    voices), exactly as in ddsp.
    """

    def __init__(self, dag, name='processor_group', fast_path=True):
        self.dag = [tuple(node) for node in dag]
        self.name = name
        self.fast_path = fast_path
        self._processors = []
        for node in self.dag:
            p = node[0]
            if not isinstance(p, Processor):
                raise TypeError(f'DAG node {node!r} does not start with a Processor')
            if all(p is not q for q in self._processors):
                self._processors.append(p)
                setattr(self, p.name, p)
        self._plan = None

    @property
    def processors(self):
        """Unique processors in first-seen DAG order (synthesize_from_csv.py:99 takes [:2])."""
        return list(self._processors)

    def _batched_plan(self):
        if self.fast_path and self._plan is None:
            from . import polyphonic
            self._plan = polyphonic.recognise(self.dag) or False
        return self._plan if self.fast_path else False

    def __call__(self, inputs, return_outputs_dict=False, **kwargs):
        # audio only: the batched route skips the voice stems; outputs dict: it adds what the reference's dict holds.  For
        # polyphonic_dag's node list that is the LAST voice's stems (polyphonic_dag.py re-uses three processor objects, so
        # every earlier voice is overwritten): need_stems='last'.  default_model.py:44-80 names every Add node (`add_i`,
        # `sub_add_i`), so the reference's dictionary holds every running sum: the complete dictionary needs every voice's
        # stems (need_stems=True); the reduced one (last pair + dry mix only) is an explicit opt-in, need_stems='last'.
        if 'need_stems' not in kwargs:
            plan = self._batched_plan()
            complete = bool(plan) and plan.shape == 'default_model'
            kwargs['need_stems'] = (True if complete else 'last') if return_outputs_dict else False
        outputs = self.get_controls(inputs, **kwargs)
        signal = self.get_signal(outputs)
        if return_outputs_dict:
            return dict(signal=signal, controls=outputs)
        return signal

    def decompose(self, inputs, noise=None):
        """The reference's --decompose flow (synthesize_from_csv.py:92-120) in one call: {'additive': the sum over the voices
        of the additive synthesiser's signals, 'noise': the sum of the noise synthesiser's, 'dry': the un-reverbed mix,
        'signal': the group's output}.  On the batched route these sums are what the compacted oscillator bank and the
        noise kernel's voice sums form anyway (need_stems='sums'); otherwise every voice's stems are rendered and added."""
        outputs = self.get_controls(inputs, noise=noise, need_stems='sums')
        sums = outputs.get('voices_sum')
        if sums is None:
            if 'voices' in outputs:                         # batched route, per-voice rows
                sums = {k: v.sum(dim=1) for k, v in outputs['voices'].items()}
            else:                                           # node-by-node walk: the reference's own loop
                # (synthesize_from_csv.py:99-120 re-runs processors[:2] once per voice, so -- as there -- the noise stems
                # are FRESH draws unless `noise` is given: 'noise' + 'additive' then differs from 'dry' by the draw.  The
                # batched route's sums are the very stems inside 'dry'.)
                from .synths import FilteredNoise, InHarmonic, SurrogateAdditive
                adds_ = [q for q in self.processors if isinstance(q, (InHarmonic, SurrogateAdditive))]
                noises_ = [q for q in self.processors if isinstance(q, FilteredNoise)]
                if len(adds_) != 1 or len(noises_) != 1:
                    raise TypeError('decompose() needs a polyphonic group: exactly one additive synthesiser (InHarmonic / '
                                    'MultiInharmonic / SurrogateAdditive) and one FilteredNoise among its processors, found '
                                    f'{[type(q).__name__ for q in self.processors]}')
                additive, noise_p = adds_[0], noises_[0]
                add_nodes = [n for n in self.dag if n[0] is additive]
                noise_nodes = [n for n in self.dag if n[0] is noise_p]
                a = z = None
                for i, node in enumerate(add_nodes):
                    x = additive.get_signal(**additive.get_controls(*[_nested_lookup(k, outputs) for k in node[1]]))
                    a = x if a is None else a + x
                for i, node in enumerate(noise_nodes):
                    kw = {}
                    if noise is not None:
                        kw['noise'] = noise[i] if isinstance(noise, (list, tuple)) else (noise[:, i] if noise.dim() == 3 else noise)
                    x = noise_p.get_signal(**noise_p.get_controls(*[_nested_lookup(k, outputs) for k in node[1]]), **kw)
                    z = x if z is None else z + x
                sums = {'additive': a, 'noise': z}
        signal = self.get_signal(outputs)
        from .effects import FeedbackDelayNetwork, FeedbackDelayNetworkApply, Reverb
        last = self.dag[-1]                                 # a reverb as the last node: its first input is the dry mix
        dry = _nested_lookup(last[1][0], outputs) if isinstance(last[0], (Reverb, FeedbackDelayNetwork,
                                                                           FeedbackDelayNetworkApply)) else signal
        return {'additive': sums['additive'], 'noise': sums['noise'], 'dry': dry, 'signal': signal}

    def get_controls(self, inputs, noise=None, need_stems=True):
        """Run the DAG; returns the full outputs dict (ddsp.dags.DAGLayer.run_dag).

        noise: explicit uniform(-1, 1) draws for the noise synthesiser instead of its own generator -- [B, P, N] or a
        sequence of P tensors [B, N], voice i = the i-th FilteredNoise node of the DAG (the reference draws them
        unseeded, filtered_noise_synth.py:39-40; parity tests and reproducible renders pass them in).
        need_stems only concerns the batched polyphonic route (polyphonic.run)."""
        plan = self._batched_plan()
        if plan:
            from . import polyphonic
            outputs = polyphonic.run(plan, inputs, noise=noise, need_stems=need_stems)
            if outputs is not None:
                return outputs
        return self._run_dag(inputs, noise=noise)

    def _run_dag(self, inputs, noise=None):
        from .synths import FilteredNoise
        outputs = {'inputs': inputs}
        outputs.update(inputs)
        module_outputs = None
        noise_calls = 0
        for node in self.dag:
            processor, input_keys = node[0], node[1]
            args = [_nested_lookup(k, outputs) for k in input_keys]
            if noise is not None and isinstance(processor, FilteredNoise):
                z = noise[noise_calls] if isinstance(noise, (list, tuple)) else \
                    (noise[:, noise_calls] if noise.dim() == 3 else noise)
                noise_calls += 1
                controls = processor.get_controls(*[core.tf_float32(a) for a in args])
                module_outputs = dict(signal=processor.get_signal(**controls, noise=z), controls=controls)
            else:
                module_outputs = processor(*args, return_outputs_dict=True)
            outputs[processor.name] = module_outputs
        outputs['out'] = module_outputs
        return outputs

    def get_signal(self, outputs):
        return outputs['out']['signal']
