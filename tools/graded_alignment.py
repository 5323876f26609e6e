#!/usr/bin/env python3
"""Does the graded kernel's time depend on how the two envelope tensors sit relative to each other in HBM?
fe and ae (37.7 GB each) are carved out of one buffer with a controlled byte offset between their (2 MiB aligned) starts."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ddsp_piano_amd as dp
from ddsp_piano_amd import core
dev = torch.device('cuda', 0)
R, N, H, sr = 1024, 72000, 128, 24000
n = R * N * H
pool = torch.empty(2 * n + (1 << 26), dtype=torch.float32, device=dev)
base = (-(pool.data_ptr() // 4)) % (1 << 19)          # to a 2 MiB boundary (in floats)
g = torch.Generator(device=dev); g.manual_seed(0)
def run(off_floats):
    fe = pool[base: base + n].view(R, N, H)
    ae = pool[base + n + off_floats: base + 2 * n + off_floats].view(R, N, H)
    fe.uniform_(100.0, 8000.0, generator=g); ae.uniform_(0.0, 0.01, generator=g)
    for _ in range(2): core.cos_oscillator_bank(fe, ae, sr, True, True, spans=1)
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); core.cos_oscillator_bank(fe, ae, sr, True, True, spans=1); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)
for off in [0, 64, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 19, (1 << 19) + 1024, 1 << 20, 3 << 19, 1 << 22, (1 << 22) + 4096]:
    mn, av = run(off)
    print(f'ae offset {off * 4:>10d} B: min {mn:7.3f} ms avg {av:7.3f} ms  ({(n * 8 + R * N * 4) / mn / 1e6:6.0f} GB/s)')
