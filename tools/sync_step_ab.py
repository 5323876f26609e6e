#!/usr/bin/env python3
"""Synchronised headline steps under option sets, interleaved in one process (same box, same clocks): for each set the median /
min / quartiles of the per-step device times.  usage: python tools/sync_step_ab.py [steps] [rounds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ddsp_piano_amd as dp  # noqa: E402
from ddsp_piano_amd import _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda', 0)
B, P, T, H, K, S, L, sr = 64, 16, 750, 128, 96, 1, 72000, 24000
feats, _ = bench.make_features(B, P, T, H, K, S, L, dev, seed=31)
pg = bench.build_group(dp, P, sr)
SETS = [('default', {}), ('DDSPP_NOISE_NO_DRAW=1', {'DDSPP_NOISE_NO_DRAW': '1'}), ('DDSPP_CONTROLS_DENSE_HD=1', {'DDSPP_CONTROLS_DENSE_HD': '1'}),
        ('both off (round 5 form)', {'DDSPP_NOISE_NO_DRAW': '1', 'DDSPP_CONTROLS_DENSE_HD': '1'})]
res = {n: [] for n, _ in SETS}
fn = lambda: pg(feats, return_outputs_dict=True)      # noqa: E731
for r in range(rounds):
    for name, env in SETS:
        for k in ('DDSPP_NOISE_NO_DRAW', 'DDSPP_CONTROLS_DENSE_HD'):
            os.environ.pop(k, None)
        os.environ.update(env)
        _lib.options.reload()
        res[name] += list(bench.event_times(fn, steps, warmup=5))
for name, _ in SETS:
    t = np.array(res[name])
    print('%-28s n %3d  median %.4f  min %.4f  q25 %.4f  q75 %.4f  max %.4f ms' % (name, len(t), np.median(t), t.min(), np.quantile(t, .25), np.quantile(t, .75), t.max()))
