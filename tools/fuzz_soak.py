#!/usr/bin/env python3
"""Randomised soak of the route cross-checks of tests/test_gpu_fuzz.py: shapes, layouts and flags drawn at random for a
given number of seconds; every failing draw is printed with its parameters.  usage: python tools/fuzz_soak.py [seconds] [seed]"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_fuzz as F  # noqa: E402
import test_gpu_native_group as G  # noqa: E402



class Env:
    """What the tests take from pytest's monkeypatch."""

    def setenv(self, k, v):
        os.environ[k] = str(v)

    def delenv(self, k, raising=False):
        os.environ.pop(k, None)


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
FLAGS = [{}, {}, dict(scale='exp_tanh', normalize_after_nyquist_cut=False), dict(normalize_below_nyquist=False),
         dict(normalize_below_nyquist=False, normalize_after_nyquist_cut=False)]
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    U = int(rng.choice([32, 64, 96, 128, 192]))
    S = int(rng.choice([1, 1, 2]))
    H = int(rng.choice([16, 48, 64, 96, 128, 192]))
    if S * H > 512:
        continue
    P = int(rng.integers(1, min(16, 64 // S) + 1))
    B = int(rng.choice([1, 2, 3, 5, 16, 20]))
    T = int(rng.integers(8, 200)) if B * P > 64 else int(rng.integers(8, 600))
    seed = int(rng.integers(1, 1 << 30))
    kind = int(rng.integers(0, 4))
    # round 5: half of the draws lower the thresholds behind which the memoised pre-pass + the compacted scan of moving chunks
    # and the two-oscillators-per-lane slots (paired sub-strings at S = 2) run, so that small random shapes take those
    # kernels too (by default they need 256 rows / 2048 wavefronts)
    forced = rng.random() < 0.5
    for k_, v_ in (('DDSPP_OSC_MEMO_MIN_WAVES', '4'), ('DDSPP_OSC_COMPACT_VPL1_BELOW', '0')):
        if forced:
            os.environ[k_] = v_
        else:
            os.environ.pop(k_, None)
    from ddsp_piano_amd import _lib as _L
    _L.options.reload()
    try:
        if kind == 0:
            args = (seed, B, P, T, H, S, U, FLAGS[int(rng.integers(0, len(FLAGS)))])
            F.test_compacted_bank_equals_the_stems(*args)
        elif kind == 1:
            K = {32: 32, 64: 64, 96: 96, 128: 96, 192: 96}[U] if rng.random() < 0.7 else int(rng.choice([32, 64, 65, 96, 128]))
            sur = S == 1 and rng.random() < 0.3          # SurrogateAdditive voices (the bank's decay variant)
            args = (seed, min(B, 5), P, min(T, 200), H, K, S, U, bool(rng.integers(0, 2)))
            F.test_batched_group_equals_the_node_by_node_walk(*args, Env(), surrogate=sur)
        elif kind == 3:                        # the one-call C driver against the Python route
            K = int(rng.choice([32, 64, 96, 128]))
            flags = None if rng.random() < 0.6 else dict(scale='exp_tanh', normalize_after_nyquist_cut=False)
            args = (seed, min(B, 5), P, max(2, min(T, 150)), H, K, S, U, bool(rng.integers(0, 2)),
                    int(rng.choice([0, 500, 3000, 9000])), flags)
            G.test_native_group_equals_the_python_route(*args, Env())
            os.environ.pop('DDSPP_VOICE_SUMS', None)
        else:
            K = {32: 32, 64: 64, 96: 96, 128: 96, 192: 96}[U]
            if U == 192:
                continue
            args = (seed, min(B, 3), min(P, 6), max(T, 260), H, K, S, U)
            F.test_streamed_pieces_equal_the_one_call_render(*args)
        n += 1
    except Exception:  # noqa: BLE001
        bad += 1
        print('FAILED', kind, args, flush=True)
        traceback.print_exc(limit=3)
print(f'fuzz soak: {n} draws passed, {bad} failed, {time.time() - t0:.0f} s (half of the draws with DDSPP_OSC_MEMO_MIN_WAVES=4, '
      'DDSPP_OSC_COMPACT_VPL1_BELOW=0: the compacted scan and the paired sub-strings on small shapes)')
