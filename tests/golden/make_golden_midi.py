#!/usr/bin/env python3
"""Golden vectors for the piano-roll -> conditioning step, produced by THE REFERENCE ITSELF.

    python tests/golden/make_golden_midi.py        (build container only: needs /root/reference)

ddsp_piano/utils/midi_encoders.py needs nothing but NumPy, so -- unlike the TensorFlow part of the
path -- the reference class can be executed here.  This script imports it from /root/reference, feeds it
seeded synthetic piano rolls and stores inputs + the reference's outputs (and its allocator state after
every call) in tests/golden/midi_conditioning.npz.  Only data is written; no reference source is copied.

Cases (all with onset velocities only on sounding keys, as note_seq's rolls have):
  sparse16     400 frames, light polyphony, n_synths=16, float64
  dense16      600 frames, up to ~40 simultaneous keys (more than the 16 channels), float64
  poly4_f32    300 frames, n_synths=4, float32 roll
  chunked8     500 frames fed in three calls to ONE object (state carries over), n_synths=8
  empty_roll   50 silent frames
"""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/ddsp_piano/utils/midi_encoders.py'


def reference_class():
    spec = importlib.util.spec_from_file_location('reference_midi_encoders', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.MIDIRoll2Conditioning


def synth_roll(rng, n_frames, n_notes, max_len):
    """(active uint8 [T, 88], velocity uint8 [T, 88]): velocity v/127 at the onset frame of each note."""
    act = np.zeros((n_frames, 88), np.uint8)
    vel = np.zeros((n_frames, 88), np.uint8)
    for _ in range(n_notes):
        key = int(rng.integers(0, 88))
        start = int(rng.integers(0, max(1, n_frames - 1)))
        stop = min(n_frames, start + int(rng.integers(1, max_len)))
        if act[max(start - 1, 0):stop, key].any():
            continue                                   # keep notes on one key apart
        act[start:stop, key] = 1
        vel[start, key] = int(rng.integers(1, 128))
    return act, vel


def dense(act, vel, dtype):
    return np.stack([act.astype(dtype), vel.astype(dtype) / dtype(127)], axis=-1)


def main():
    Ref = reference_class()
    rng = np.random.default_rng(20240928)
    out = {}
    cases = [('sparse16', 400, 40, 80, 16, np.float64, [400]),
             ('dense16', 600, 700, 120, 16, np.float64, [600]),
             ('poly4_f32', 300, 60, 60, 4, np.float32, [300]),
             ('chunked8', 500, 150, 90, 8, np.float64, [137, 1, 362]),
             ('empty_roll', 50, 0, 10, 16, np.float64, [50])]
    for name, T, notes, max_len, n_synths, dtype, chunks in cases:
        act, vel = synth_roll(rng, T, notes, max_len)
        enc = Ref(n_synths)
        conds, polys, states = [], [], []
        at = 0
        for n in chunks:
            part = dense(act[at:at + n], vel[at:at + n], dtype)
            c, p = enc(part)                            # the reference scales `part` in place; it is a temporary
            conds.append(c)
            polys.append(p)
            states.append(np.concatenate([[enc.assigner], enc.reorder, enc.assigned_pitch]))
            at += n
        out[f'{name}/active'] = act
        out[f'{name}/velocity'] = vel
        out[f'{name}/n_synths'] = np.int64(n_synths)
        out[f'{name}/float32'] = np.bool_(dtype is np.float32)
        out[f'{name}/chunks'] = np.asarray(chunks)
        out[f'{name}/conditioning'] = np.concatenate(conds, axis=0)
        out[f'{name}/polyphony'] = np.concatenate(polys, axis=0)
        out[f'{name}/states'] = np.stack(states)         # per call: [assigner, reorder[n], assigned_pitch[n]]
        print(name, out[f'{name}/conditioning'].shape, 'max polyphony', int(out[f'{name}/polyphony'].max()))
    np.savez_compressed(os.path.join(HERE, 'midi_conditioning.npz'), **out)


if __name__ == '__main__':
    main()
