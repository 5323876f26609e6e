"""Reverb processors: ddsp.effects.Reverb (maestro-v2.gin:152-164, default_model.py:77-80) and the
*apply* step of FeedbackDelayNetwork (ddsp_piano/modules/fdn_reverb.py:407-410).  The FDN impulse
response *generation* (fdn_reverb.py:178-360) is control-side and out of scope."""
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
