#!/usr/bin/env python3
"""The batched group's FilteredNoise call alone on the bench inputs (for rocprofv3): noise_case.py [vq] [reps]."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ddsp_piano_amd as dp  # noqa: E402
from ddsp_piano_amd import core  # noqa: E402

vq = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda', 0)
B, P, T, H, K, S = 64, 16, 750, 128, 96, 1
feats, base = bench.make_features(B, P, T, H, K, S, 72000, dev, 1)
noise = dp.DynamicSizeFilteredNoise(sample_rate=24000)
mags = base['magnitudes'].reshape(B * P, T, K)
x = core.uniform_noise((B * P, T * 96), seed=1, device=dev)
rs = noise.raw_scale()
for _ in range(reps):
    if vq > 1:
        core.frequency_filter_voice_sums(x, mags, noise.window_size, rs, P, vq, False, split_last=True)
    else:
        core.frequency_filter(x, mags, window_size=noise.window_size, raw_scale=rs)
torch.cuda.synchronize()
