import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench, ddsp_piano_amd as dp
from ddsp_piano_amd import core
dev = torch.device('cuda', 0)
B, P, T, H, K, S, sr, L = 32, 32, 750, 128, 96, 1, 48000, 480000
N = T * 192
feats, base = bench.make_features(B, P, T, H, K, S, L, dev, seed=5)
R = B * P
add = dp.MultiInharmonic(sample_rate=sr, inference=True)
ctl = add._controls(base['amplitudes'].reshape(R, T, 1), base['harmonic_distribution'].reshape(R, T, H),
                    base['inharm_coef'].reshape(R, T, 1), base['f0_hz'].reshape(R, T, S), want_counts=True)
torch.cuda.synchronize(); print('controls ok', flush=True)
for sp in (1, 0):
    for rows in (64, 1024):
        try:
            y = core.harmonic_synthesis_fused(ctl['f0_hz'][:rows], ctl['amplitudes'].reshape(R, T)[:rows], ctl['harmonic_distribution'][:rows],
                                              ctl['harmonic_shifts'][:rows], N, sr, True, spans=sp)
            torch.cuda.synchronize(); print('stems spans', sp, 'rows', rows, 'ok', float(y.abs().max()), flush=True)
        except Exception as e:
            print('stems spans', sp, 'rows', rows, 'FAILED', str(e)[:200], flush=True); raise
y = core.polyphonic_additive(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T), ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, sr,
                             audible=ctl['_audible'], split_last=True)
torch.cuda.synchronize(); print('compact ok', flush=True)
