#!/usr/bin/env python3
"""The headline batch through ddspp_group_run (NativeGroup) a few times, for rocprofv3 --kernel-trace --stats:
usage: python tools/trace_native.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ddsp_piano_amd as dp  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda', 0)
B, P, T, H, K, S, L, sr = 64, 16, 750, 128, 96, 1, 72000, 24000
feats, _ = bench.make_features(B, P, T, H, K, S, L, dev, seed=31)
pg = bench.build_group(dp, P, sr)
ng = dp.NativeGroup(pg, feats)
ts = bench.event_times(lambda: ng(feats, return_outputs_dict=True), reps, warmup=3)
print('native dict ms per step:', bench.ms_summary(ts))
ts = bench.event_times(lambda: pg(feats, return_outputs_dict=True), reps, warmup=3)
print('python dict ms per step:', bench.ms_summary(ts))
