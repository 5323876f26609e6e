#!/usr/bin/env python3
"""Run one bench input case a few times (for rocprofv3 --kernel-trace --stats).
usage: python tools/trace_case.py [headline|midi|moving|dense|c5|dafx22|dafx24|multi|surrogate|enst8k|enst32k|file|one] [audio|dict|stems] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ddsp_piano_amd as dp  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else 'headline'
form = sys.argv[2] if len(sys.argv) > 2 else 'dict'
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device('cuda', 0)
B, P, T, H, K, S, L, sr = 64, 16, 750, 128, 96, 1, 72000, 24000
if case == 'c5':            # BASELINE config 5's per-GPU share (bench.py: c5_per_gpu_share)
    B, P, H, K, S, sr, L = 32, 32, 128, 96, 1, 48000, 480000
elif case == 'dafx22':      # configs/dafx22.gin dims (bench.py: dafx22_dims)
    B, P, H, K, S, sr, L = 64, 16, 96, 64, 2, 16000, 24000
elif case in ('dafx24', 'dafx24moving'):      # configs/dafx22-24kHz.gin dims: two sub-strings at 24 kHz
    B, P, H, K, S, sr, L = 64, 16, 128, 96, 2, 24000, 36000
elif case in ('multi', 'multimoving'):       # configs/multi_instruments.gin dims
    B, P, H, K, S, sr, L = 64, 16, 96, 64, 1, 16000, 24000
elif case == 'surrogate':   # configs/surrogate.gin dims and flags (bench.py: shipped_configs)
    B, P, H, K, S, sr, L = 64, 16, 96, 64, 1, 16000, 16000
elif case == 'file':        # bench.py whole_file: 136 s as one segment
    B, T, L = 1, 34000, 48000
elif case == 'one':         # bench.py single_stream: one 3 s segment
    B = 1
elif case == 'enst8k':      # configs/ENSTDkCl-8kHz.gin dims (here with ddsp.effects.Reverb as the last node)
    B, P, H, K, S, sr, L = 64, 16, 48, 32, 1, 8000, 16000
elif case in ('enst32k', 'enst32kmoving'):     # configs/ENSTDkCl-32kHz.gin dims
    B, P, H, K, S, sr, L = 64, 16, 192, 128, 1, 32000, 64000
kw = {'headline': {}, 'midi': {}, 'moving': dict(vibrato=0.002), 'dafx24moving': dict(vibrato=0.002), 'multimoving': dict(vibrato=0.002), 'enst32kmoving': dict(vibrato=0.002), 'c5': {}, 'dafx22': {}, 'dafx24': {}, 'multi': {}, 'surrogate': {}, 'enst8k': {}, 'enst32k': {}, 'file': {}, 'one': {},
      'dense': dict(silent_frac=0.0, midi_lo=21, midi_hi=33, vibrato=0.004)}[case]
if case == 'midi':          # bench.py midi_like: note-shaped controls from a synthetic piano roll
    feats, _, _st = bench.make_midi_like_features(dp, B, P, T, H, K, S, L, dev, seed=33)
    print('midi_like inputs:', _st)
else:
    feats, _ = bench.make_features(B, P, T, H, K, S, L, dev, seed=31, **kw)
pg = bench.build_group(dp, P, sr)
if case == 'surrogate':
    g = torch.Generator(device=dev)
    g.manual_seed(44)
    dec = 0.9990 + 0.0012 * torch.rand(B, P, T, H, generator=g, device=dev)
    dt = torch.arange(T, device=dev, dtype=torch.float32).view(1, 1, T, 1).expand(B, P, T, 1).contiguous()
    for i in range(P):
        feats[f'decays_{i}'], feats[f'decay_time_{i}'] = dec[:, i], dt[:, i]
    pg = bench.build_shipped_group(dp, 'surrogate', P, sr)
fn = {'dict': lambda: pg(feats, return_outputs_dict=True), 'audio': lambda: pg(feats),
      'stems': lambda: pg(feats, return_outputs_dict=True, need_stems=True)}[form]
ts = bench.event_times(fn, reps, warmup=2)
print(case, form, 'ms per step:', bench.ms_summary(ts))
