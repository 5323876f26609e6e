"""Merging / un-merging of the batch and polyphony axes on the input edge of the synthesis path.

Mirror of ``Parallelizer`` (ddsp_piano/modules/sub_modules.py:527-602): the control networks run with the
polyphony folded into the batch ([P * B, T, C], voice major) and ``unparallelize`` hands the ProcessorGroup
one key per voice, ``<key>_<i>`` = rows [i * B, (i + 1) * B).  Here those per-voice entries are VIEWS of the
merged buffer (no 5 * P copies); ``ProcessorGroup`` recognises them and runs its kernels straight on the
merged buffer (polyphonic._stack_voices, voice-major rows).
"""
from __future__ import annotations

import torch


class Parallelizer:
    """Args as the reference's: n_synths, global_keys (merged on the way in), mono_keys (split per voice on
    the way out)."""

    def __init__(self, n_synths=16,
                 global_keys=('conditioning', 'context', 'global_inharm', 'global_detuning'),
                 mono_keys=('f0_hz', 'inharm_coef', 'amplitudes', 'harmonic_distribution', 'magnitudes')):
        self.n_synths = int(n_synths)
        self.global_keys = tuple(global_keys)
        self.mono_keys = tuple(mono_keys)
        self.batch_size = None

    def build(self, features):
        self.batch_size = int(features['conditioning'].shape[0])          # sub_modules.py:553-555

    def put_polyphony_axis_at_first(self, x):
        """[B, T] / [B, T, C] -> shared by all voices [P, B, ...]; [B, T, P, C] -> [P, B, T, C]  (:557-568)."""
        if 2 <= x.dim() <= 3:
            return x.unsqueeze(0).expand((self.n_synths,) + tuple(x.shape))
        if x.dim() == 4:
            return x.permute(2, 0, 1, 3)
        return x

    def parallelize_feature(self, x):
        """[P, B, ...] -> [P * B, ...]  (:570-575)."""
        return x.reshape((self.n_synths * self.batch_size,) + tuple(x.shape[2:]))

    def unparallelize_feature(self, x):
        """[P * B, ...] -> [P, B, ...]  (:577-582); a view."""
        return x.reshape((self.n_synths, self.batch_size) + tuple(x.shape[1:]))

    def parallelize(self, features):
        for k in self.global_keys:
            features[k] = self.parallelize_feature(self.put_polyphony_axis_at_first(features[k]))
        return features

    def unparallelize(self, features):
        """features[k] becomes [P, B, T, C] and features[k + '_i'] its i-th voice (:584-592)."""
        for k in self.mono_keys:
            features[k] = self.unparallelize_feature(features[k])
            for i in range(self.n_synths):
                features[f'{k}_{i}'] = features[k][i]
        return features

    def __call__(self, features, parallelize=True):
        if self.batch_size is None:
            self.build(features)
        return self.parallelize(features) if parallelize else self.unparallelize(features)
