"""Two half batches on two streams against one call of the group (profiles/r03_ubench.txt): is there anything to gain from
overlapping the memory-bound kernels of one half with the issue-bound kernels of the other?  (No: 5 % slower.)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, ddsp_piano_amd as dp
dev = torch.device('cuda', 0)
B, P, T, H, K, S, L, sr = 64, 16, 750, 128, 96, 1, 72000, 24000
feats, _ = bench.make_features(B, P, T, H, K, S, L, dev, seed=31)
pg = bench.build_group(dp, P, sr)
def half(f, lo, hi):
    return {k: v[lo:hi] for k, v in f.items()}
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fs = [half(feats, i * B // parts, (i + 1) * B // parts) for i in range(parts)]
pgs = [bench.build_group(dp, P, sr) for _ in range(parts)]
streams = [torch.cuda.Stream(dev) for _ in range(parts)]
def split_call():
    cur = torch.cuda.current_stream(dev)
    outs = []
    for s, g, f in zip(streams, pgs, fs):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(g(f, return_outputs_dict=True)['signal'])
    for s in streams:
        cur.wait_stream(s)
    return outs
full = lambda: pg(feats, return_outputs_dict=True)['signal']
for name, fn in (('full', full), (f'{parts} parts on {parts} streams', split_call), ('full', full), (f'{parts} parts', split_call)):
    ts = bench.event_times(fn, 20, warmup=3)
    print(name, bench.ms_summary(ts))
