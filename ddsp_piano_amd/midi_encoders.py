"""Input edge of the synthesis path: piano roll -> polyphonic conditioning (host side).

Mirror of ddsp_piano/utils/midi_encoders.py:4-104 (``MIDIRoll2Conditioning``) and of the roll handling in
ddsp_piano/utils/io_utils.py:113-137, :204-224 -- same class / function names, arguments and return values.
The frame-sequential voice allocator runs in the native library (``ddspp_midi_conditioning_*``,
csrc/midi_conditioning.cpp) on HOST buffers: it is loop-carried bookkeeping over 88 keys per 4 ms frame, not
GPU work.  MIDI file parsing (note_seq) is outside the path and not rebuilt.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


class MIDIRoll2Conditioning(object):
    """Convert piano rolls into the polyphonic conditioning vector (midi_encoders.py:4-22).

    Params:
        n_synths (int): supported number of simultaneous notes.
    Attributes ``assigner``, ``reorder``, ``assigned_pitch`` and ``pitch_mul`` read like the reference
    object's; the allocator state persists from one call to the next, as the reference object's does.
    """

    def __init__(self, n_synths=16):
        self.n_synths = int(n_synths)
        self.pitch_mul = np.arange(21, 21 + 88)
        self._lib = _lib.load()
        self._handle = self._lib.ddspp_midi_conditioning_create(self.n_synths)
        if not self._handle:
            raise ValueError(self._lib.ddspp_last_error().decode())

    def __del__(self):
        handle, self._handle = getattr(self, '_handle', None), None
        if handle:
            self._lib.ddspp_midi_conditioning_destroy(handle)

    def reset(self):
        """Back to the state of a freshly constructed object."""
        _lib.check(self._lib.ddspp_midi_conditioning_reset(self._handle))

    def _state(self):
        assigner = ctypes.c_int(0)
        reorder = np.zeros(self.n_synths, dtype=np.int32)
        pitch = np.zeros(self.n_synths, dtype=np.float64)
        _lib.check(self._lib.ddspp_midi_conditioning_get_state(
            self._handle, ctypes.addressof(assigner), reorder.ctypes.data, pitch.ctypes.data))
        return assigner.value, reorder.astype(int), pitch

    @property
    def assigner(self):
        return self._state()[0]

    @property
    def reorder(self):
        return self._state()[1]

    @property
    def assigned_pitch(self):
        return self._state()[2]

    def __call__(self, roll):
        """roll (n_frames, 88, 2): stacked active and onset-velocity piano rolls ->
        (conditioning (n_frames, n_synths, 2), polyphony (n_frames,))   -- midi_encoders.py:33-104.

        Unlike the reference, ``roll`` is left untouched (the reference scales its activity plane by the
        pitch in place, :49)."""
        roll = np.asarray(roll)
        if roll.ndim != 3 or roll.shape[2] != 2:
            raise ValueError(f'roll must be (n_frames, 88, 2), got {roll.shape}')
        dtype = np.float32 if roll.dtype == np.float32 else np.float64
        roll = np.ascontiguousarray(roll, dtype=dtype)
        n_frames, n_pitches = roll.shape[:2]
        cond = np.empty((n_frames, self.n_synths, 2), dtype=dtype)
        poly = np.empty((n_frames,), dtype=dtype)
        fn = self._lib.ddspp_midi_conditioning_run_f32 if dtype == np.float32 else \
            self._lib.ddspp_midi_conditioning_run_f64
        _lib.check(fn(self._handle, roll.ctypes.data, n_frames, n_pitches, cond.ctypes.data, poly.ctypes.data))
        return cond, poly


def ensure_sequence_length(sequence, length, right=True):
    """Zero-pad or crop ``sequence`` (time, ...) to ``length`` frames, at its end (right=True) or its
    beginning -- io_utils.py:204-224."""
    sequence = np.asarray(sequence)
    length = int(length)
    have = sequence.shape[0]
    if have == length:
        return sequence
    if have > length:
        return sequence[:length] if right else sequence[have - length:]
    pad = [(0, length - have) if right else (length - have, 0)] + [(0, 0)] * (sequence.ndim - 1)
    return np.pad(sequence, pad_width=pad)


def roll_to_conditioning(active, onset_velocities, control_changes, total_time, n_synths=16, frame_rate=250,
                         duration=None, warm_up_duration=0.):
    """What load_midi_as_conditioning (io_utils.py:91-137) does once note_seq has produced the piano roll:
    active / onset_velocities (n_frames, 88), control_changes (n_frames, 128), total_time = the note
    sequence's length in seconds.  Returns {'conditioning' (1, n, n_synths, 2), 'pedal' (1, n, 4), 'duration'}."""
    midi_roll = np.stack((np.asarray(active), np.asarray(onset_velocities)), axis=-1)
    pedals = np.asarray(control_changes)[:, 64:68] / 128.
    conditioning, _ = MIDIRoll2Conditioning(n_synths)(midi_roll)
    if duration is None:
        target = int(np.ceil(total_time) * frame_rate)
    else:
        target = int(duration * frame_rate)
    conditioning = ensure_sequence_length(conditioning, target)
    pedals = ensure_sequence_length(pedals, target)
    if warm_up_duration > 0.:
        n_frames = target + int(warm_up_duration * frame_rate)
        conditioning = ensure_sequence_length(conditioning, n_frames, right=False)
        pedals = ensure_sequence_length(pedals, n_frames, right=False)
    return {'conditioning': conditioning[np.newaxis, ...], 'pedal': pedals[np.newaxis, ...],
            'duration': target / frame_rate + warm_up_duration}
