import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import ddsp_piano_amd as dp
from ddsp_piano_amd import core
from tools.bench_kernels import timeit
dev = torch.device('cuda', 0)
B, P, T, H, K, S = 64, 16, 750, 128, 96, 1
N = T * 96
feats, base = bench.make_features(B, P, T, H, K, S, 72000, dev, 1)
R = B * P
additive = dp.MultiInharmonic(sample_rate=24000, inference=True)
ctl = additive._controls(base['amplitudes'].reshape(R, T, 1), base['harmonic_distribution'].reshape(R, T, H),
                         base['inharm_coef'].reshape(R, T, 1), base['f0_hz'].reshape(R, T, S))
ha = (ctl['amplitudes'] * ctl['harmonic_distribution'])
nact = (ha != 0).any(dim=1).sum(dim=1).float()
print('mean audible harmonics per voice', nact.mean().item(), 'per segment', nact.reshape(B, P).sum(1).mean().item())
for sp in (4, 8, 16, 24, 36, 72):
    fn = lambda: core.polyphonic_additive(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T), ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, 24000, spans=sp)
    mn, av = timeit(fn, 3)
    print('compact spans', sp, 'min %.3f ms' % mn)
a = core.polyphonic_additive(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T), ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, 24000)
b = core.harmonic_synthesis_fused(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T), ctl['harmonic_distribution'], ctl['harmonic_shifts'], N, 24000, True).reshape(B, P, N).sum(1)
print('max diff', (a - b).abs().max().item(), 'rms', b.pow(2).mean().sqrt().item())
