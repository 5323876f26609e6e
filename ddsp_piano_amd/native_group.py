"""The polyphonic group through the library's one-call driver (csrc/group.cpp, ``ddspp_group_*`` in include/ddspp.h).

``ProcessorGroup`` strings the kernels together from Python (polyphonic.py): a dozen ctypes calls, torch allocations,
table look-ups -- 0.25 ms of host time per call, hidden behind the GPU at batch 64, exposed for one segment.
``NativeGroup`` does the same sequence inside the library: one call, one workspace.  Same kernels, same arguments: the
results equal the batched Python route's to the last bits (the two Hann table builders differ by an ulp in a few
entries; tests/test_gpu_native_group.py).  It takes the polyphonic_dag
shape with the library's own scale functions and, as its last node, a ddsp.effects.Reverb, a FeedbackDelayNetwork that holds its
parameters (the ENSTDkCl configurations) or nothing; anything else stays with ProcessorGroup.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib, core
from .core import _lib_, _ptr, _stream
from .effects import FeedbackDelayNetwork, FeedbackDelayNetworkApply, Reverb
from .polyphonic import _stack_voices, noise_rows, recognise
from .synths import InHarmonic


class _Config(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ('n_segments', 'n_voices', 'n_frames', 'n_harmonics', 'n_substrings', 'n_bands',
                                            'upsampling', 'ir_length', 'ir_batch', 'reverb_add_dry', 'voice_major')] + \
               [('sample_rate', ctypes.c_float), ('min_frequency', ctypes.c_float), ('scale_kind', ctypes.c_int)] + \
               [(n, ctypes.c_float) for n in ('exponent', 'max_value', 'threshold', 'gain')] + \
               [('normalize_after_nyquist_cut', ctypes.c_int), ('normalize_below_nyquist', ctypes.c_int),
                ('window_size', ctypes.c_int), ('noise_scale_kind', ctypes.c_int)] + \
               [(n, ctypes.c_float) for n in ('noise_bias', 'noise_exponent', 'noise_max_value', 'noise_threshold',
                                              'noise_gain')] + \
               [('delay_compensation', ctypes.c_int), ('resize_rule', ctypes.c_int), ('noise_seed', ctypes.c_uint64),
                ('reverb_keep_dry_tap', ctypes.c_int), ('reserved_', ctypes.c_int)]


class _Outputs(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ('dry', 'prev', 'additive_last', 'noise_last', 'amplitudes_last',
                                               'harmonic_distribution_last', 'harmonic_shifts_last', 'magnitudes_last')]


class NativeGroup:
    """group = NativeGroup(processor_group, example_features); audio = group(features) / group(features,
    return_outputs_dict=True, noise=...) -- the call forms of ProcessorGroup, for inputs of the example's shapes."""

    def __init__(self, group, features):
        plan = recognise(group.dag)
        if plan is None or plan.shape != 'gin':
            raise ValueError('NativeGroup takes the node list of polyphonic_dag(...) over this package\'s processors')
        add, nz, rv = plan.additive, plan.noise, plan.reverb
        if not isinstance(add, InHarmonic) or not add.inference:
            raise ValueError('NativeGroup needs the inference oscillator (angular cumsum)')
        # the last node: ddsp.effects.Reverb with the impulse response as a control; a FeedbackDelayNetwork that holds its
        # parameters (reverb_controls = [], configs/ENSTDkCl-8kHz.gin:85-104: its impulse response is computed once, here);
        # the apply step of one whose impulse response is a control; or nothing
        self._fdn = rv is not None and isinstance(rv, (FeedbackDelayNetwork, FeedbackDelayNetworkApply))
        if self._fdn:
            if isinstance(rv, FeedbackDelayNetwork) and not (rv.trainable and len(plan.reverb_keys) == 0):
                raise ValueError('NativeGroup takes a FeedbackDelayNetwork that holds its parameters (trainable=True, reverb_controls=[])')
            if isinstance(rv, FeedbackDelayNetworkApply) and len(plan.reverb_keys) != 1:
                raise ValueError('FeedbackDelayNetworkApply needs its impulse response as the one reverb control')
        elif rv is not None and (type(rv) is not Reverb or len(plan.reverb_keys) != 1):
            raise ValueError('NativeGroup takes ddsp.effects.Reverb with the impulse response as a control, a FeedbackDelayNetwork '
                             'that holds its parameters, or no reverb')
        ak, zk = core.scale_kind(add.scale_fn), (nz.raw_scale() if nz.scale_fn is not None else (-1, 0.0, core.scale_kind(None)[1]))
        if ak is None or zk is None:
            raise ValueError('NativeGroup needs the library\'s scale functions (exp_sigmoid, exp_tanh, None)')
        self.plan = plan
        self.dag = group.dag                 # (CapturedGroup looks the noise processors up here)
        P = plan.n_synths
        hd0 = features[plan.additive_keys[0][1]]
        B, T, H = hd0.shape
        S = features[plan.additive_keys[0][3]].shape[-1]
        K = features[plan.noise_keys[0]].shape[-1]
        self._vm = self._layout(features)[5]
        c = _Config()
        c.n_segments, c.n_voices, c.n_frames, c.n_harmonics, c.n_substrings, c.n_bands = B, P, T, H, S, K
        c.upsampling = add.upsampling
        ir = self._impulse_response(features, hd0.device)
        c.ir_length = int(ir.shape[-1]) if ir is not None else 0
        c.ir_batch = (1 if (ir.dim() == 1 or ir.shape[0] == 1) else B) if ir is not None else 0
        c.reverb_add_dry = 0 if self._fdn else (int(rv._add_dry) if rv is not None else 1)
        c.reverb_keep_dry_tap = 1 if self._fdn else 0
        c.voice_major = int(self._vm)
        c.sample_rate, c.min_frequency = float(add.sample_rate), float(add.min_frequency)
        c.scale_kind = ak[0]
        c.exponent, c.max_value, c.threshold, c.gain = (float(ak[1][k]) for k in ('exponent', 'max_value', 'threshold', 'gain'))
        c.normalize_after_nyquist_cut = int(add.normalize_after_nyquist_cut)
        c.normalize_below_nyquist = int(add.normalize_below_nyquist)
        c.window_size = int(nz.window_size)
        c.noise_scale_kind, c.noise_bias = int(zk[0]), float(zk[1])
        c.noise_exponent, c.noise_max_value, c.noise_threshold, c.noise_gain = \
            (float(zk[2][k]) for k in ('exponent', 'max_value', 'threshold', 'gain'))
        c.delay_compensation = core._auto_delay(-1)
        c.resize_rule = 1 if core.RECALLED['resize'] == 'half_pixel' else 0
        c.noise_seed = int(getattr(nz, 'seed', 0)) & (2 ** 64 - 1)
        self.config = c
        lib = _lib_()
        if ctypes.sizeof(_Config) != lib.ddspp_group_config_bytes() or ctypes.sizeof(_Outputs) != lib.ddspp_group_outputs_bytes():
            raise RuntimeError('ddspp_group_config / ddspp_group_outputs: this binding and libddspp.so disagree on the layout')
        h = ctypes.c_void_p()
        _lib.check(lib.ddspp_group_create(ctypes.byref(c), ctypes.byref(h)))
        self._h = h
        self._lib = lib
        self.N = int(lib.ddspp_group_n_samples(h))
        self._ws_bytes = int(lib.ddspp_group_workspace_bytes(h))
        self._ws = torch.empty(self._ws_bytes, dtype=torch.uint8, device=hd0.device)
        self.dims = (B, P, T, H, S, K)

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                self._lib.ddspp_group_destroy(h)
            except Exception:  # noqa: BLE001  (interpreter teardown)
                pass

    def _impulse_response(self, features, dev):
        plan = self.plan
        rv = plan.reverb
        if rv is None:
            return None
        if isinstance(rv, FeedbackDelayNetwork):        # parameters held by the layer (it keeps the impulse response until
            return rv.get_controls(torch.empty(0, device=dev))['ir']       # load_parameters replaces them): one for all rows
        return features[plan.reverb_keys[0]]

    def _layout(self, features):
        plan = self.plan
        ctl = [[features[k[j]] for k in plan.additive_keys] for j in range(4)]
        hd, vm = _stack_voices(ctl[1], getattr(self, '_vm', None))
        amp, _ = _stack_voices(ctl[0], vm)
        inh, _ = _stack_voices(ctl[2], vm)
        f0, _ = _stack_voices(ctl[3], vm)
        mags, _ = _stack_voices([features[k] for k in plan.noise_keys], vm)
        return amp, hd, inh, f0, mags, vm

    def __call__(self, features, return_outputs_dict=False, noise=None):
        B, P, T, H, S, K = self.dims
        N = self.N
        amp, hd, inh, f0, mags, vm = self._layout(features)
        if tuple(hd.shape) != (B * P, T, H) or tuple(mags.shape) != (B * P, T, K) or f0.shape[-1] != S:
            raise ValueError('features do not have the shapes this NativeGroup was created for')
        dev = hd.device
        plan = self.plan
        ir = self._impulse_response(features, dev)
        if ir is not None:
            ir = core.tf_float32(ir)
            ir = (ir[None, :] if ir.dim() == 1 else ir).contiguous()
        z = noise_rows(noise, B, P, N, vm) if noise is not None else None
        audio = torch.empty((B, N), dtype=torch.float32, device=dev)
        outs, o = None, None
        if return_outputs_dict:
            def new(*shape):
                return torch.empty(shape, dtype=torch.float32, device=dev)
            outs = dict(dry=new(B, N), prev=new(B, N) if P > 1 else None, additive_last=new(B, N), noise_last=new(B, N),
                        amplitudes_last=new(B, T, 1), harmonic_distribution_last=new(B, T, H),
                        harmonic_shifts_last=new(B, T, H), magnitudes_last=new(B, T, K))
            o = _Outputs(**{k: (v.data_ptr() if v is not None else None) for k, v in outs.items()})
        _lib.check(self._lib.ddspp_group_run(self._h, _ptr(amp), _ptr(hd), _ptr(inh), _ptr(f0), _ptr(mags), _ptr(ir),
                                             _ptr(z), _ptr(audio), ctypes.byref(o) if o is not None else None,
                                             _ptr(self._ws), self._ws_bytes, _stream()))
        if not return_outputs_dict:
            return audio
        last = P - 1

        def voice(x, shape):
            return (x.reshape((P, B) + shape).transpose(0, 1) if vm else x.reshape((B, P) + shape))[:, last]
        add, nz, mix = plan.additive, plan.noise, plan.add
        shifts = outs['harmonic_shifts_last']
        outputs = {'inputs': features}
        outputs.update(features)
        outputs[add.name] = {'signal': outs['additive_last'],
                             'controls': {'amplitudes': outs['amplitudes_last'],
                                          'harmonic_distribution': outs['harmonic_distribution_last'],
                                          'harmonic_shifts': shifts, 'f0_hz': voice(f0, (T, S))}}
        outputs[nz.name] = {'signal': outs['noise_last'], 'controls': {'magnitudes': outs['magnitudes_last']}}
        addc = {'signal_0': outs['prev'], 'signal_1': outs['noise_last'], 'signal_2': outs['additive_last']} if P > 1 else \
            {'signal_0': outs['noise_last'], 'signal_1': outs['additive_last']}
        outputs[mix.name] = {'signal': outs['dry'], 'controls': addc}
        module = outputs[mix.name]
        if plan.reverb is not None:
            ir_ctl = self._impulse_response(features, dev)
            module = {'signal': audio, 'controls': {'audio': outs['dry'], 'ir': ir_ctl}}
            outputs[plan.reverb.name] = module
        outputs['out'] = module
        return {'signal': module['signal'], 'controls': outputs}
