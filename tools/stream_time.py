#!/usr/bin/env python3
"""Latency of the streaming synthesiser: one poly-16 voice set at 24 kHz pushed in blocks of `frames` control frames
(125 frames = 0.5 s of audio).  usage: python tools/stream_time.py [frames per push] [pushes]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ddsp_piano_amd as dp  # noqa: E402
from ddsp_piano_amd import streaming  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 125
pushes = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device('cuda', 0)
B, P, H, K, S, sr, L = 1, 16, 128, 96, 1, 24000, 48000
T = frames * (pushes + 4)
feats, _ = bench.make_features(B, P, T, H, K, S, L, dev, seed=5)
syn = streaming.StreamingSynthesizer(dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True),
                                     dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr),
                                     dp.Reverb(name='reverb'), n_synths=P)
keys = [k for k in feats if k != 'reverb_ir']


def piece(i):
    f = {k: feats[k][:, i * frames:(i + 1) * frames] for k in keys}
    f['reverb_ir'] = feats['reverb_ir']
    return f


for i in range(4):
    syn.push(piece(i))
torch.cuda.synchronize()
ts = []
for i in range(4, 4 + pushes):
    t0 = time.perf_counter()
    out = syn.push(piece(i))
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ms = float(np.median(ts)) * 1e3
print(f'push of {frames} frames ({frames / 250:.2f} s of audio): median {ms:.3f} ms, min {min(ts) * 1e3:.3f} ms '
      f'-> {frames / 250 / (ms * 1e-3):.0f} x real time; last output {tuple(out.shape)}')
