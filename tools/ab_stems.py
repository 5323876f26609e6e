#!/usr/bin/env python3
"""Every voice's additive stem, two ways, on the headline inputs: the per-voice fused kernel (ddspp_harmonic_synthesis, what
need_stems=True runs) against the compacted bank with every voice as its own segment (ddspp_polyphonic_additive, B' = B P,
P' = 1).  and against ddspp_polyphonic_stems (the voices of a segment packed, the harmonic sum stopped at voice boundaries).
usage: python tools/ab_stems.py [case]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ddsp_piano_amd as dp  # noqa: E402
from ddsp_piano_amd import core, _lib, polyphonic  # noqa: E402

dev = torch.device('cuda', 0)
B, P, T, H, K, S, L, sr = 64, 16, 750, 128, 96, 1, 72000, 24000
case = sys.argv[1] if len(sys.argv) > 1 else 'headline'
if case == 'dafx24':
    S, L = 2, 36000
kw = dict(vibrato=0.002) if case == 'moving' else {}
feats, _ = bench.make_features(B, P, T, H, K, S, L, dev, seed=31, **kw)
U = sr // 250
N = T * U
R = B * P
syn = dp.MultiInharmonic(sample_rate=sr, inference=True)
keys = ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')
rows = {k: polyphonic._stack_voices([feats[f'{k}_{i}'] for i in range(P)])[0] for k in keys}
ctl_a = syn._controls(rows['amplitudes'], rows['harmonic_distribution'], rows['inharm_coef'], rows['f0_hz'], want_counts=False, want_shifts=True)
ctl_b = syn._controls(rows['amplitudes'], rows['harmonic_distribution'], rows['inharm_coef'], rows['f0_hz'], want_counts=True, want_shifts=False)


def fused():
    return core.harmonic_synthesis_fused(ctl_a['f0_hz'], ctl_a['amplitudes'].reshape(R, T), ctl_a['harmonic_distribution'],
                                         ctl_a['harmonic_shifts'], N, sr, True)


def bank():
    return core.polyphonic_additive(ctl_b['f0_hz'], ctl_b['amplitudes'].reshape(R, T), ctl_b['harmonic_distribution'], None, R, N, sr,
                                    audible=ctl_b['_audible'], inharm_coef=ctl_b['_inharm_coef'].reshape(R, T))


def packed():
    return core.polyphonic_stems(ctl_b['f0_hz'], ctl_b['amplitudes'].reshape(R, T), ctl_b['harmonic_distribution'], None, B, N, sr,
                                 audible=ctl_b['_audible'], inharm_coef=ctl_b['_inharm_coef'].reshape(R, T))


a, b, c = fused(), bank(), packed()
print(case, 'max |fused - bank| =', (a - b).abs().max().item(), 'max |fused - packed| =', (a - c).abs().max().item(),
      'rms', a.pow(2).mean().sqrt().item())
for name, fn, opts in (('fused rows', fused, {}), ('packed stems (ddspp_polyphonic_stems)', packed, {}), ('bank P=1 (64-oscillator slots)', bank, {}),
                       ('bank P=1, 128-oscillator slots', bank, {'DDSPP_OSC_COMPACT_VPL1_SINGLE': 0})):
    for k, v in opts.items():
        _lib.set_option(k, v)
    ts = bench.event_times(fn, 10, warmup=3)
    print(f'{name:40s}', bench.ms_summary(ts))
    for k in opts:
        _lib.set_option(k, 1)
