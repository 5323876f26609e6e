#!/usr/bin/env python3
"""Does the graded kernel's rate depend on WHERE its two envelope buffers lie?  One process, one 80 GB allocation; fe at
offset 0, ae behind it with a pad; the same kernel on the same data for a list of pads (and once with two separate
allocations).  usage: python tools/osc_placement.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ddsp_piano_amd as dp  # noqa: E402
from ddsp_piano_amd import core  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda', 0)
R, N, H, sr = 1024, 72000, 128, 24000
n = R * N * H
g = torch.Generator(device=dev); g.manual_seed(1)
max_pad = 64 << 20
pool = torch.empty(2 * n * 4 + max_pad + 4096, dtype=torch.uint8, device=dev)
fe = pool[:n * 4].view(torch.float32).view(R, N, H)
fe.copy_(torch.rand(R, 1, H, device=dev, generator=g) * 4000.0 + 20.0)
print('pool at', hex(pool.data_ptr()))


def t(fe, ae):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); core.cos_oscillator_bank(fe, ae, sr, True, True, spans=1); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


byts = R * (N * H * 8 + N * 4)
for pad in (0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 32 << 20, (32 << 20) + 16384):
    ae = pool[n * 4 + pad:n * 4 + pad + n * 4].view(torch.float32).view(R, N, H)
    ae.fill_(1e-3)
    ms = t(fe, ae)
    print(f'pad {pad:9d} B: {ms:7.3f} ms  {byts / ms / 1e6:7.1f} GB/s')
del pool, fe, ae
torch.cuda.empty_cache()
fe = torch.rand(R, 1, H, device=dev).expand(R, N, H).contiguous() * 4000.0
ae = torch.full((R, N, H), 1e-3, device=dev)
ms = t(fe, ae)
print(f'two allocations {hex(fe.data_ptr())} {hex(ae.data_ptr())}: {ms:7.3f} ms  {byts / ms / 1e6:7.1f} GB/s')
