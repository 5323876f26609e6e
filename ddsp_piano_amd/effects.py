"""Reverb processors: ddsp.effects.Reverb (maestro-v2.gin:152-164, default_model.py:77-80) and the
FeedbackDelayNetwork (ddsp_piano/modules/fdn_reverb.py): impulse-response generation (:178-360),
the apply step (:407-410), and the parameter-holding form the ENSTDkCl configs use as a DAG node."""
from __future__ import annotations

import torch

from . import core
from .processors import Processor


class Reverb(Processor):
    """ddsp.effects.Reverb(trainable=False, reverb_length=48000, add_dry=True, name='reverb')."""

    def __init__(self, trainable=False, reverb_length=48000, add_dry=True, name='reverb'):
        super().__init__(name=name, trainable=trainable)
        self._reverb_length = reverb_length
        self._add_dry = add_dry
        self._ir = None
        if trainable:
            # ddsp initialises N(0, 1e-6); training the IR is out of scope, but the inference
            # behaviour of a trainable Reverb (own IR, tiled over the batch) is kept.
            self._ir = torch.zeros(int(reverb_length), dtype=torch.float32)

    def _match_dimensions(self, audio, ir):
        """Tile the ir to match the batch of the audio (ddsp.effects.Reverb._match_dimensions)."""
        if ir.dim() == 1:
            ir = ir[None, :]
        if ir.dim() == 3:
            ir = ir[:, :, 0]
        return ir

    def get_controls(self, audio, ir=None):
        if self.trainable:
            ir = core.tf_float32(self._ir, device=core.tf_float32(audio).device)[None, :]
        else:
            if ir is None:
                raise ValueError('Must provide "ir" tensor if Reverb trainable=False.')
        return {'audio': audio, 'ir': ir}

    def get_signal(self, audio, ir):
        audio, ir = core.tf_float32(audio), core.tf_float32(ir)
        ir = self._match_dimensions(audio, ir).contiguous()
        if audio.dim() != 2:
            raise ValueError('audio must be [batch, n_samples]')
        if ir.shape[0] != audio.shape[0] and ir.shape[0] != 1:
            raise ValueError('Batch size of audio ({}) and impulse response ({}) must be the same.'.format(
                audio.shape[0], ir.shape[0]))
        # _mask_dry_ir + fft_convolve(padding='same', delay_compensation=0) + dry, one rocFFT pipeline
        return core._fft_convolve_single(audio, ir, 'same', 0, mask_dry=True, add_dry=self._add_dry)


    # The impulse response is an input of the group, known before the dry mix exists: the batched group transforms it
    # early (begin, on its side stream) and convolves when the mix is there (finish).  Same kernels as get_signal.
    def begin(self, batch, n_samples, ir, key_stream=None):
        if self.trainable or ir is None:
            return None
        ir = core.tf_float32(ir)
        ir = self._match_dimensions(None, ir).contiguous()
        if ir.dim() != 2 or (ir.shape[0] != batch and ir.shape[0] != 1):
            return None
        return core.fft_convolve_prepare(batch, n_samples, ir, mask_dry=True, key_stream=key_stream)

    def finish(self, state, audio):
        return core.fft_convolve_finish(state, audio, 'same', 0, add_dry=self._add_dry)


FDN_DELAY_VALUES = (233., 311., 421., 461., 587., 613., 789., 891.)                      # fdn_reverb.py:96
FDN_DELAYS_ALLPASS = ((131., 151., 337., 353.), (103., 173., 331., 373.), (89., 181., 307., 401.),
                      (79., 197., 281., 419.), (61., 211., 257., 431.), (47., 229., 251., 443.),
                      (81., 189., 287., 407.), (91., 203., 321., 377.))                  # fdn_reverb.py:102-113

_irfft_plans = core._PlanCache('ddspp_irfft_plan_destroy', maxsize=8)


def _irfft(spectrum, n):
    """tf.signal.irfft for [B, n/2+1] complex64 -> [B, n] float32 (rocFFT C2R; the spectrum buffer is consumed)."""
    import ctypes
    from . import _lib
    from .core import _lib_, _ptr, _stream
    b = spectrum.shape[0]

    def create():
        handle = ctypes.c_void_p()
        with torch.cuda.device(spectrum.device):
            _lib.check(_lib_().ddspp_irfft_plan_create(n, b, ctypes.byref(handle)))
        return handle
    entry = _irfft_plans.get((n, b, str(spectrum.device)), create, pin=True)
    try:
        plan = entry[0]
        nbytes = int(_lib_().ddspp_irfft_workspace_bytes(plan))
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=spectrum.device)
        out = torch.empty((b, n), dtype=torch.float32, device=spectrum.device)
        with entry[1]:
            _lib.check(_lib_().ddspp_irfft_execute(plan, _ptr(spectrum), _ptr(out), _ptr(ws), nbytes, _stream()))
    finally:
        _irfft_plans.pin(entry, -1)
    return out


def fdn_impulse_response(input_gain, output_gain, gain_allpass, delays_allpass, time_rev_0_sec, alpha_tone,
                         early_ir=None, delay_values=FDN_DELAY_VALUES, sampling_rate=16000.0, mixing_matrix=None):
    """FeedbackDelayNetwork.get_ir for B instruments at once (fdn_reverb.py:178-360; the
    tf.vectorized_map of MultiInstrumentFeedbackDelayReverb.call, sub_modules.py:431-446).

    input_gain / output_gain [B, D]; gain_allpass / delays_allpass [B, D, A]; time_rev_0_sec, alpha_tone
    [B] (or [B, 1]); early_ir [B, E] or None  ->  ir [B, 2 * sampling_rate].
    core.RECALLED['fdn_solve'] selects the float64 solve (default) or the reference's complex64 inverse.
    """
    from . import _lib
    from .core import _lib_, _ptr, _stream
    input_gain, output_gain = core.tf_float32(input_gain), core.tf_float32(output_gain)
    gain_allpass, delays_allpass = core.tf_float32(gain_allpass), core.tf_float32(delays_allpass)
    if input_gain.dim() != 2 or gain_allpass.dim() != 3:
        raise ValueError('input_gain must be [batch, delay_lines] and gain_allpass [batch, delay_lines, stages]')
    b, d = input_gain.shape
    a = gain_allpass.shape[-1]
    dev = input_gain.device
    t0 = core.tf_float32(time_rev_0_sec, device=dev).reshape(b).contiguous()
    al = core.tf_float32(alpha_tone, device=dev).reshape(b).contiguous()
    dv = core.tf_float32(torch.as_tensor(delay_values, dtype=torch.float32), device=dev).contiguous()
    if dv.numel() != d:
        raise ValueError(f'{dv.numel()} delay values for {d} delay lines')
    if mixing_matrix is None:
        mixing_matrix = -1.0 * torch.eye(d) + 0.5 * torch.ones(d, d)                    # fdn_reverb.py:118-120
    mix = core.tf_float32(mixing_matrix, device=dev)
    freq_points = int(2 * float(sampling_rate))                                       # fdn_reverb.py:81
    nb = freq_points // 2 + 1
    spec = torch.empty((b, nb, 2), dtype=torch.float32, device=dev)
    _lib.check(_lib_().ddspp_fdn_transfer(_ptr(input_gain), _ptr(output_gain), _ptr(mix), _ptr(gain_allpass),
                                          _ptr(delays_allpass), _ptr(t0), _ptr(al), _ptr(dv), _ptr(spec), b, d, a,
                                          freq_points, float(sampling_rate),
                                          1 if core.RECALLED['fdn_solve'] == 'complex64' else 0, _stream()))
    ir = _irfft(spec, freq_points)
    if early_ir is not None:
        early = core.tf_float32(early_ir, device=dev).reshape(b, -1).contiguous()
        _lib.check(_lib_().ddspp_fdn_add_early(_ptr(ir), _ptr(early), b, freq_points, early.shape[1], _stream()))
    return ir


class FeedbackDelayNetwork(Processor):
    """Frequency-sampled feedback delay network reverb, ddsp_piano/modules/fdn_reverb.py:20-410.

    trainable=False: every parameter arrives through get_controls (the reference's non-trainable branch, :387-396).
    trainable=True: the layer holds its parameters -- the reference's weights of build() (:130-176): early_ir
    [early_ir_length], input_gain / output_gain [D], time_rev_0_sec, alpha_tone (scalars; sigmoid applied in
    get_controls), delays_allpass / gain_allpass [D, 4] and, with delay_trainable, delay_values [D].  This package
    never trains them: they are initialised with the reference's initialisers (seeded) or set from a checkpoint with
    load_parameters().  That makes the layer usable as the LAST node of the polyphonic DAG with reverb_controls = []
    (configs/ENSTDkCl-8kHz.gin:85-86,100-104, ENSTDkCl-32kHz.gin).  The impulse response of fixed parameters is
    computed once and kept until load_parameters() is called again (cache_ir=True, the default: weights do not move at
    inference); cache_ir=False designs it inside every get_controls, as the reference does (fdn_reverb.py:383-392) --
    the like-for-like cost of a call (bench.py reports both)."""

    PARAMETER_NAMES = ('early_ir', 'input_gain', 'output_gain', 'time_rev_0_sec', 'alpha_tone', 'delay_values',
                       'delays_allpass', 'gain_allpass')

    def __init__(self, trainable=False, name='DelayNetwork', sampling_rate=16000.0, delay_lines=8,
                 delay_values=None, delays_allpass=None, early_ir_length=200, early_reflections=6,
                 time_control_bands=6, delay_trainable=False, seed=0, cache_ir=True):
        super().__init__(name=name, trainable=trainable)
        self.cache_ir = bool(cache_ir)
        self.sampling_rate = float(sampling_rate)
        self.freq_points = int(2 * self.sampling_rate)                  # :81
        self.early_ir_length = early_ir_length
        self.early_reflections = early_reflections
        self.time_control_bands = time_control_bands
        self.delay_trainable = delay_trainable
        self._ir_cache = None
        self._dev_params = None
        if trainable:
            d = int(delay_lines)
            g = torch.Generator().manual_seed(int(seed))

            def normal(mean, std, *shape):
                return (torch.randn(*shape, generator=g) * std + mean).to(torch.float32)
            # the initialisers of build() (:130-176)
            if delay_values is not None:
                dv = torch.as_tensor(delay_values, dtype=torch.float32)
                d = dv.numel()
            elif delay_trainable:
                dv = normal(400.0, 60.0, d)
            else:
                dv = torch.tensor(FDN_DELAY_VALUES, dtype=torch.float32)
                d = dv.numel()
            self._params = {
                'early_ir': normal(0.0, 0.1, int(early_ir_length)),
                'input_gain': normal(0.25, 0.1, d),
                'output_gain': normal(0.25, 0.1, d),
                'time_rev_0_sec': normal(2.0, 0.5, 1).abs().reshape(()),      # NonNeg constraint
                'alpha_tone': normal(0.0, 0.1, 1).reshape(()),
                'delay_values': dv,
                'delays_allpass': normal(400.0, 60.0, d, 4),
                'gain_allpass': normal(0.25, 0.1, d, 4),
            }
            self.delay_values = tuple(float(v) for v in dv)
            self.delays_allpass = None
        else:
            self._params = None
            self.delay_values = tuple(delay_values) if delay_values is not None else FDN_DELAY_VALUES
            self.delays_allpass = delays_allpass if delays_allpass is not None else FDN_DELAYS_ALLPASS
        self.delay_lines = len(self.delay_values)

    def __len__(self):
        return self.delay_lines

    def parameters(self):
        """The layer's own parameters (trainable=True), by the reference's weight roles."""
        if self._params is None:
            raise ValueError('a FeedbackDelayNetwork with trainable=False holds no parameters')
        return dict(self._params)

    def load_parameters(self, params):
        """Set the layer's parameters (e.g. read from a checkpoint of the reference): a mapping with any of
        PARAMETER_NAMES.  Shapes follow build(): [D], [D, A], scalars, [early_ir_length]."""
        if self._params is None:
            raise ValueError('a FeedbackDelayNetwork with trainable=False holds no parameters')
        new = dict(self._params)
        for k, v in params.items():
            if k not in self.PARAMETER_NAMES:
                raise KeyError(f'unknown FeedbackDelayNetwork parameter {k!r}; known: {self.PARAMETER_NAMES}')
            new[k] = torch.as_tensor(v, dtype=torch.float32).detach().cpu().clone()
        d = new['delay_values'].numel()
        for k in ('input_gain', 'output_gain'):
            if new[k].numel() != d:
                raise ValueError(f'{k} has {new[k].numel()} entries for {d} delay lines')
        for k in ('delays_allpass', 'gain_allpass'):
            if new[k].dim() != 2 or new[k].shape[0] != d or new[k].shape != new['gain_allpass'].shape:
                raise ValueError(f'{k} must be [{d}, stages], got {tuple(new[k].shape)}')
        self._params = new
        self.delay_values = tuple(float(v) for v in new['delay_values'])
        self.delay_lines = d
        self._ir_cache = None
        self._dev_params = None

    def get_ir(self, input_gain, output_gain, gain_allpass, delays_allpass, time_rev_0_sec, alpha_tone, early_ir):
        """fdn_reverb.py:336-360 (one instrument; returns [2 * sampling_rate])."""
        def one(x, nd):
            x = core.tf_float32(x)
            return x.reshape((1,) + tuple(x.shape[-nd:])) if nd else x.reshape(1)
        ir = fdn_impulse_response(one(input_gain, 1), one(output_gain, 1), one(gain_allpass, 2),
                                  one(delays_allpass, 2), one(time_rev_0_sec, 0), one(alpha_tone, 0),
                                  one(torch.as_tensor(early_ir).reshape(-1), 1), self.delay_values,
                                  self.sampling_rate)
        return ir[0]

    def get_controls(self, audio_dry=None, input_gain=None, output_gain=None, gain_allpass=None,
                     delays_allpass=None, time_rev_0_sec=None, alpha_tone=None, early_ir=None):
        """fdn_reverb.py:362-405."""
        if self.trainable:                                                         # :383-392
            dev = core.tf_float32(audio_dry).device if audio_dry is not None else core.default_device()
            if self._ir_cache is None or self._ir_cache.device != dev or not self.cache_ir:
                if self._dev_params is None or self._dev_params[0] != dev:
                    self._dev_params = (dev, {k: v.to(dev) for k, v in self._params.items()})
                p = self._dev_params[1]
                self._ir_cache = self.get_ir(p['input_gain'], p['output_gain'], p['gain_allpass'], p['delays_allpass'],
                                             p['time_rev_0_sec'], torch.sigmoid(p['alpha_tone']), p['early_ir'])
            return {'audio': audio_dry, 'ir': self._ir_cache}
        ir = self.get_ir(input_gain, output_gain, gain_allpass, delays_allpass, time_rev_0_sec, alpha_tone, early_ir)
        return {'audio': audio_dry, 'ir': ir}

    def get_signal(self, audio, ir):
        """fdn_reverb.py:407-410."""
        ir = core.tf_float32(ir)
        return core.fft_convolve(audio, ir[None, :], delay_compensation=0)


class FeedbackDelayNetworkApply(Processor):
    """The get_signal half of FeedbackDelayNetwork (fdn_reverb.py:407-410): the controls already
    carry a finished impulse response ``ir [L]``; no dry mask, no dry add."""

    def __init__(self, name='fdn_reverb'):
        super().__init__(name=name)

    def get_controls(self, audio, ir):
        return {'audio': audio, 'ir': ir}

    def get_signal(self, audio, ir):
        ir = core.tf_float32(ir)
        if ir.dim() != 1:
            raise ValueError('FeedbackDelayNetwork impulse response must be 1-D [ir_size]')
        return core.fft_convolve(audio, ir[None, :], delay_compensation=0)


class _InstrumentTables:
    """tf.keras.layers.Embedding tables of a multi-instrument reverb layer: name -> [n_instruments, width] float32,
    kept on the host and moved to the device of the first call (then cached)."""

    def __init__(self, tables):
        self._host = {k: torch.as_tensor(v, dtype=torch.float32).contiguous() for k, v in tables.items()}
        self._dev = {}

    def names(self):
        return tuple(self._host)

    def host(self):
        return dict(self._host)

    def load(self, params, n_instruments):
        new = dict(self._host)
        for k, v in params.items():
            if k not in new:
                raise KeyError(f'unknown embedding table {k!r}; known: {tuple(new)}')
            v = torch.as_tensor(v, dtype=torch.float32).detach().cpu().clone()
            if v.shape != new[k].shape:
                raise ValueError(f'{k} must be {tuple(new[k].shape)} ([n_instruments={n_instruments}, width]), '
                                 f'got {tuple(v.shape)}')
            new[k] = v.contiguous()
        self._host, self._dev = new, {}

    def lookup(self, name, index):
        dev = index.device
        key = (name, str(dev))
        if key not in self._dev:
            self._dev[key] = self._host[name].to(dev)
        return self._dev[key].index_select(0, index)


def _instrument_index(piano_model, n_instruments):
    """`piano_model` [B, 1] (or [B]) integer ids -> [B] int64 on its device; n_instruments == 1 maps every id to 0
    (sub_modules.py:353-354, :432-433)."""
    idx = torch.as_tensor(piano_model)
    if not idx.is_cuda:
        idx = idx.to(core.default_device())
    if idx.dim() == 2:
        idx = idx[..., 0]
    if idx.dim() != 1:
        raise ValueError('piano_model must be [batch, 1] instrument ids')
    idx = idx.to(torch.int64)
    return torch.zeros_like(idx) if n_instruments == 1 else idx


class MultiInstrumentReverb:
    """ddsp_piano/modules/sub_modules.py:302-365: one learnt impulse response per instrument (an Embedding of
    reverb_length values, initialised N(0, 1e-6)); at inference an exponential decay mask past sample 16000.
    `layer(features)` reads features['piano_model'] [B, 1] and returns {'reverb_ir': [B, reverb_length]} (the
    nn.DictLayer contract PianoModel relies on, piano_model.py:99-125); `layer.call(piano_model)` returns the tensor."""

    def __init__(self, n_instruments=16, reverb_duration=1.5, sample_rate=16000, inference=False, seed=0,
                 name='multi_instrument_reverb'):
        self.name = name
        self.n_instruments = int(n_instruments)
        self.reverb_duration = reverb_duration
        self.sample_rate = sample_rate
        self.inference = inference
        g = torch.Generator().manual_seed(int(seed))
        self._tables = _InstrumentTables(
            {'reverb_dict': torch.randn(self.n_instruments, self.reverb_length, generator=g) * 1e-6})

    @property
    def reverb_length(self):
        return int(self.reverb_duration * self.sample_rate)

    def parameters(self):
        return self._tables.host()

    def load_parameters(self, params):
        """{'reverb_dict': [n_instruments, reverb_length]} -- e.g. the dafx22 checkpoint's reverb bank."""
        self._tables.load(params, self.n_instruments)

    def exponential_decay_mask(self, ir, decay_exponent=4., decay_start=16000):
        """sub_modules.py:339-349."""
        n = self.reverb_length - decay_start
        time = torch.linspace(0.0, 1.0, n, dtype=torch.float32, device=ir.device)
        mask = torch.cat([torch.ones(decay_start, dtype=torch.float32, device=ir.device),
                          torch.exp(-decay_exponent * time)])
        return ir * mask[None, :]

    def call(self, piano_model):
        ir = self._tables.lookup('reverb_dict', _instrument_index(piano_model, self.n_instruments))
        return self.exponential_decay_mask(ir) if self.inference else ir

    def __call__(self, features, training=False):
        pm = features['piano_model'] if isinstance(features, dict) else features
        return {'reverb_ir': self.call(pm)}


class MultiInstrumentFeedbackDelayReverb:
    """ddsp_piano/modules/sub_modules.py:368-446 -- how the default (maestro-v2) configuration produces `reverb_ir`
    (configs/maestro-v2.gin:118-122): seven Embedding tables hold one feedback-delay network per instrument,
    `call(piano_model [B, 1]) -> reverb_ir [B, 2 * sample_rate]` looks the B parameter sets up, conditions two of them
    (relu on time_rev_0_sec, sigmoid on alpha_tone, :440-441), and evaluates FeedbackDelayNetwork.get_ir for all of them
    at once (the reference's tf.vectorized_map, :444; here one launch of ddspp_fdn_transfer + one batched C2R).

    Table layout (the Embedding weights, rows = instruments): input_gain / output_gain [n, D]; gain_allpass /
    delays_allpass [n, 4 D] with reshape_embedding's split-then-stack order (:427-429) -- column a D + d of the table is
    all-pass stage a of delay line d, i.e. the [D, 4] matrix is table.reshape(4, D).T, NOT table.reshape(D, 4);
    time_rev_0_sec / alpha_tone [n, 1]; early_ir [n, early_ir_length].  Initialisers as :386-418 (seeded);
    load_parameters() takes a checkpoint's arrays."""

    TABLES = ('input_gain', 'output_gain', 'gain_allpass', 'delays_allpass', 'time_rev_0_sec', 'alpha_tone', 'early_ir')

    def __init__(self, n_instruments=10, sample_rate=16000, delay_lines=8, early_ir_length=200, regularize_early=False,
                 seed=0, name='multi_instrument_feedback_delay_reverb'):
        self.name = name
        self.n_instruments = int(n_instruments)
        self.sample_rate = sample_rate
        self.delay_lines = int(delay_lines)
        self.early_ir_length = int(early_ir_length)
        self.regularize_early = regularize_early            # a training-time L1 penalty: nothing to do at synthesis
        n, d = self.n_instruments, self.delay_lines
        g = torch.Generator().manual_seed(int(seed))

        def normal(mean, std, *shape):
            return (torch.randn(*shape, generator=g) * std + mean).to(torch.float32)
        self._tables = _InstrumentTables({
            'input_gain': normal(0.25, 0.1, n, d), 'output_gain': normal(0.25, 0.1, n, d),
            'gain_allpass': normal(0.25, 0.1, n, 4 * d), 'delays_allpass': normal(400.0, 60.0, n, 4 * d),
            'time_rev_0_sec': normal(2.0, 0.5, n, 1), 'alpha_tone': normal(0.0, 0.1, n, 1),
            'early_ir': normal(0.0, 0.1, n, self.early_ir_length)})
        # the reference's inner FeedbackDelayNetwork(trainable=False, sampling_rate=sample_rate): default delay lines
        self.reverb_model = FeedbackDelayNetwork(trainable=False, sampling_rate=self.sample_rate)
        if self.reverb_model.delay_lines != d:
            raise ValueError(f'the inner FeedbackDelayNetwork has {self.reverb_model.delay_lines} delay lines '
                             f'(fdn_reverb.py:96), delay_lines={d} cannot be evaluated (neither can the reference)')

    def parameters(self):
        return self._tables.host()

    def load_parameters(self, params):
        self._tables.load(params, self.n_instruments)

    @staticmethod
    def reshape_embedding(embedding, splits=4):
        """sub_modules.py:427-429: tf.stack(tf.split(embedding, splits, axis=-1), axis=-1): [..., splits * D] ->
        [..., D, splits] with out[..., d, a] = embedding[..., a * D + d]."""
        d = embedding.shape[-1] // splits
        return embedding.reshape(embedding.shape[:-1] + (splits, d)).transpose(-1, -2).contiguous()

    def controls(self, piano_model):
        """The controls_dict of :434-443, batched over the B looked-up instruments."""
        idx = _instrument_index(piano_model, self.n_instruments)
        t = self._tables
        return {'input_gain': t.lookup('input_gain', idx), 'output_gain': t.lookup('output_gain', idx),
                'gain_allpass': self.reshape_embedding(t.lookup('gain_allpass', idx)),
                'delays_allpass': self.reshape_embedding(t.lookup('delays_allpass', idx)),
                'time_rev_0_sec': torch.relu(t.lookup('time_rev_0_sec', idx)),
                'alpha_tone': torch.sigmoid(t.lookup('alpha_tone', idx)),
                'early_ir': t.lookup('early_ir', idx)}

    def call(self, piano_model):
        c = self.controls(piano_model)
        return fdn_impulse_response(c['input_gain'], c['output_gain'], c['gain_allpass'], c['delays_allpass'],
                                    c['time_rev_0_sec'], c['alpha_tone'], c['early_ir'],
                                    self.reverb_model.delay_values, self.reverb_model.sampling_rate)

    def __call__(self, features, training=False):
        pm = features['piano_model'] if isinstance(features, dict) else features
        return {'reverb_ir': self.call(pm)}
