#!/usr/bin/env python3
"""HIP-event time of ddspp_polyphonic_additive (flags pre-pass + counts + compacted bank + slot sum) on a bench case.
usage: python tools/bank_time.py [headline|moving|dense] [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ddsp_piano_amd as dp  # noqa: E402
from ddsp_piano_amd import core  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else 'headline'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda', 0)
B, P, T, H, K, S, L, sr = 64, 16, 750, 128, 96, 1, 72000, 24000
kw = {'headline': {}, 'moving': dict(vibrato=0.002),
      'dense': dict(silent_frac=0.0, midi_lo=21, midi_hi=33, vibrato=0.004)}[case]
_, base = bench.make_features(B, P, T, H, K, S, L, dev, seed=20240, **kw)
R, N = B * P, T * 96
add = dp.MultiInharmonic(sample_rate=sr, inference=True)
from_inh = os.environ.get('FROM_SHIFTS') != '1'          # the batched group's route: shifts formed in the kernels from inharm_coef
ctl = add._controls(base['amplitudes'].reshape(R, T, 1), base['harmonic_distribution'].reshape(R, T, H),
                    base['inharm_coef'].reshape(R, T, 1), base['f0_hz'].reshape(R, T, S), want_counts=True,
                    want_shifts=not from_inh)
split = os.environ.get('SPLIT_LAST') == '1'
fn = lambda: core.polyphonic_additive(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T), ctl['harmonic_distribution'],  # noqa: E731
                                      None if from_inh else ctl['harmonic_shifts'], B, N, sr, audible=ctl['_audible'],
                                      inharm_coef=ctl['_inharm_coef'].reshape(R, T) if from_inh else None, split_last=split)
ts = bench.event_times(fn, reps, warmup=3)
print(f'{case} ablate={os.environ.get("DDSPP_BANK_ABLATE", "0")} split={int(split)}: polyphonic_additive '
      f'median {np.median(ts):.3f} ms min {np.min(ts):.3f} ms')
