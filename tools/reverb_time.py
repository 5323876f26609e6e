#!/usr/bin/env python3
"""HIP-event time of the reverb node alone (ddsp.effects.Reverb.get_signal: dry-masked IR, + dry) at a bench shape.
usage: python tools/reverb_time.py [B N L reps]   (defaults: 64 72000 72000 20)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ddsp_piano_amd as dp  # noqa: E402

B, N, L, reps = [int(a) for a in sys.argv[1:5]] + [64, 72000, 72000, 20][len(sys.argv) - 1:]
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(1)
audio = torch.randn(B, N, generator=g, device=dev) * 0.1
ir = torch.randn(B, L, generator=g, device=dev) * torch.exp(-6.9 * torch.arange(L, device=dev) / L) * 0.02
rv = dp.Reverb()
ts = bench.event_times(lambda: rv.get_signal(audio, ir), reps, warmup=3)
alg = (N + L + N) * 4 * B
print(f'reverb B={B} N={N} L={L} partitioned={os.environ.get("DDSPP_FFT_PARTITIONED", "1")}: median {np.median(ts):.4f} ms min {np.min(ts):.4f} ms '
      f'({alg / 1e6:.0f} MB algorithmic)')
