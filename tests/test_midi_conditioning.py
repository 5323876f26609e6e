"""CPU: piano roll -> polyphonic conditioning, the one step of the chain whose oracle is PINNED.

tests/golden/midi_conditioning.npz holds inputs and outputs of the reference's own MIDIRoll2Conditioning
(ddsp_piano/utils/midi_encoders.py), run in the build container by tests/golden/make_golden_midi.py.
Checked bit-exact against it: the oracle restatement and the native host function behind the C-ABI
(ddspp_midi_conditioning_*), including the allocator state after every call.
"""
import ctypes
import os

import numpy as np
import pytest

from util import O

from ddsp_piano_amd import _lib
from ddsp_piano_amd import midi_encoders as M

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'midi_conditioning.npz'))
CASES = sorted({k.split('/')[0] for k in GOLD.files})


def _roll(name):
    dtype = np.float32 if bool(GOLD[f'{name}/float32']) else np.float64
    act, vel = GOLD[f'{name}/active'], GOLD[f'{name}/velocity']
    return np.stack([act.astype(dtype), vel.astype(dtype) / dtype(127)], axis=-1)


def _state(enc):
    return np.concatenate([[enc.assigner], np.asarray(enc.reorder), np.asarray(enc.assigned_pitch)])


@pytest.mark.parametrize('impl', ['oracle', 'native'])
@pytest.mark.parametrize('name', CASES)
def test_matches_the_reference_run(name, impl):
    roll = _roll(name)
    enc = (O if impl == 'oracle' else M).MIDIRoll2Conditioning(int(GOLD[f'{name}/n_synths']))
    at = 0
    for i, n in enumerate(GOLD[f'{name}/chunks']):
        before = roll[at:at + n].copy()
        cond, poly = enc(roll[at:at + n])
        want_c, want_p = GOLD[f'{name}/conditioning'][at:at + n], GOLD[f'{name}/polyphony'][at:at + n]
        assert cond.dtype == want_c.dtype and poly.dtype == want_p.dtype
        assert np.array_equal(cond, want_c) and np.array_equal(poly, want_p), (name, impl, i)
        assert np.array_equal(_state(enc), GOLD[f'{name}/states'][i]), (name, impl, i)
        assert np.array_equal(roll[at:at + n], before)             # the caller's roll is left alone
        at += n


def _random_roll(rng, T, notes, max_len, dtype, fractional):
    act = np.zeros((T, 88), dtype)
    vel = np.zeros((T, 88), dtype)
    for _ in range(notes):
        k, s = int(rng.integers(0, 88)), int(rng.integers(0, max(1, T - 1)))
        e = min(T, s + int(rng.integers(1, max_len)))
        act[s:e, k] = rng.uniform(0.2, 1.0) if fractional else 1.0
        vel[s, k] = rng.integers(1, 128) / 127.
    return np.stack([act, vel], axis=-1)


def test_native_equals_oracle_on_random_rolls():
    rng = np.random.default_rng(99)
    for trial in range(40):
        T, n = int(rng.integers(1, 300)), int(rng.choice([1, 2, 5, 16, 32, 88]))
        dtype = [np.float64, np.float32][trial % 2]
        roll = _random_roll(rng, T, int(rng.integers(0, 6 * n + 20)), int(rng.integers(2, 150)), dtype, trial % 5 == 4)
        a, b = O.MIDIRoll2Conditioning(n), M.MIDIRoll2Conditioning(n)
        cut = int(rng.integers(0, T + 1))
        for part in (roll[:cut], roll[cut:]):
            ca, pa = a(part)
            cb, pb = b(part)
            assert ca.shape == cb.shape == (len(part), n, 2) and ca.dtype == cb.dtype == dtype
            assert np.array_equal(ca, cb) and np.array_equal(pa, pb), (trial, T, n)
            assert np.array_equal(_state(a), _state(b)), (trial, T, n)


def test_properties_of_the_allocation():
    """A sounding note never changes channel, and every frame's channels hold the top-n pitches exactly once."""
    rng = np.random.default_rng(5)
    roll = _random_roll(rng, 800, 300, 100, np.float64, False)
    cond, poly = M.MIDIRoll2Conditioning(16)(roll)
    pitches = cond[..., 0]
    keys_down = [set(np.nonzero(roll[t, :, 0])[0] + 21) for t in range(len(roll))]
    for t in range(len(roll)):
        sounding = pitches[t][pitches[t] > 0]
        assert len(set(sounding)) == len(sounding)
        assert set(sounding) == set(sorted(keys_down[t])[-16:])
        assert poly[t] == len(keys_down[t])
        if t:
            for ch in range(16):                       # a pitch present in both frames stays where it was
                if pitches[t - 1, ch] > 0 and pitches[t - 1, ch] in sounding:
                    assert pitches[t, ch] == pitches[t - 1, ch]
    on = roll[..., 1] > 0                              # every onset velocity reaches the note's channel
    for t, k in zip(*np.nonzero(on)):
        if (k + 21) in pitches[t]:
            assert cond[t, list(pitches[t]).index(k + 21), 1] == roll[t, k, 1]


def test_edge_cases_and_errors():
    enc = M.MIDIRoll2Conditioning(16)
    cond, poly = enc(np.zeros((0, 88, 2)))
    assert cond.shape == (0, 16, 2) and poly.shape == (0,)
    assert enc.assigner == 0 and list(enc.reorder) == list(range(16)) and not enc.assigned_pitch.any()
    with pytest.raises(ValueError):
        enc(np.zeros((4, 87, 2)))                      # the reference's pitch table has 88 entries
    with pytest.raises(ValueError):
        enc(np.zeros((4, 88)))
    for bad in (0, 89, -1):
        with pytest.raises(ValueError):
            M.MIDIRoll2Conditioning(bad)
    lib = _lib.load()
    assert lib.ddspp_midi_conditioning_run_f64(None, None, 1, 88, None, None) == _lib.DDSPP_EINVAL
    enc(np.ones((3, 88, 2)))                           # every key down: the 16 highest win
    assert sorted(enc.assigned_pitch) == list(range(93, 109)) and enc.assigner == -1
    enc.reset()
    assert enc.assigner == 0 and not enc.assigned_pitch.any()
    ints = np.zeros((5, 88, 2), dtype=np.int64)        # integer rolls are computed in float64
    ints[1:4, 39, 0] = 1
    cond, _ = enc(ints)
    assert cond.dtype == np.float64 and cond[2, 0, 0] == 60.0


def test_sequence_length_and_roll_to_conditioning():
    x = np.arange(12.0).reshape(6, 2)
    for length in (6, 4, 9):
        for right in (True, False):
            got, want = M.ensure_sequence_length(x, length, right), O.ensure_sequence_length(x, length, right)
            assert got.shape == (length, 2) and np.array_equal(got, want)
    assert np.array_equal(M.ensure_sequence_length(x, 4, right=False), x[2:])
    assert np.array_equal(M.ensure_sequence_length(x, 8, right=False)[:2], np.zeros((2, 2)))
    rng = np.random.default_rng(1)
    roll = _random_roll(rng, 730, 40, 90, np.float32, False)
    cc = np.zeros((730, 128), np.float32)
    cc[100:300, 64] = 127
    out = M.roll_to_conditioning(roll[..., 0], roll[..., 1], cc, total_time=2.92, n_synths=16, frame_rate=250,
                                 warm_up_duration=0.5)
    assert out['conditioning'].shape == (1, 750 + 125, 16, 2) and out['pedal'].shape == (1, 875, 4)
    assert out['duration'] == 3.5 and out['pedal'][0, 125 + 150, 0] == np.float32(127 / 128)
    assert not out['conditioning'][0, :125].any()
    want, _ = O.MIDIRoll2Conditioning(16)(roll)
    assert np.array_equal(out['conditioning'][0, 125:125 + 730], want)
    short = M.roll_to_conditioning(roll[..., 0], roll[..., 1], cc, total_time=2.92, duration=1.0)
    assert short['conditioning'].shape == (1, 250, 16, 2) and short['duration'] == 1.0
