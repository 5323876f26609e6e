#!/usr/bin/env python3
"""Does bench.py's graded-kernel figure depend on WHEN in the process it is measured?  The same measure_roofline() in a
fresh process (nothing allocated before) and after the heap churn of a bench run's other sections.
usage: python tools/roofline_order.py [fresh|late]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ddsp_piano_amd as dp  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'fresh'
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device('cuda', 0)
B, P, T, H, K, S, L, sr = 64, 16, 750, 128, 96, 1, 72000, 24000
U = sr // 250
feats, base = bench.make_features(B, P, T, H, K, S, L, dev, seed=31)
if mode == 'late':                      # what precedes the figure in bench.py: steps, a 60 GB-ish churn of buffers
    pg = bench.build_group(dp, P, sr)
    for _ in range(5):
        pg(feats, return_outputs_dict=True)
    for gb in (20, 40, 10, 30):
        x = torch.empty(gb << 28, dtype=torch.float32, device=dev)
        x.fill_(1.0)
        del x
        torch.cuda.empty_cache()
del feats
torch.cuda.empty_cache()
r = bench.measure_roofline(dp, base, args, T, U, dev)
print(mode, 'frac', round(r['frac'], 4), 'ms', round(r['ms_per_launch'], 3), 'probe', round(r['measured_peak']))
