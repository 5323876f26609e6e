import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import ddsp_piano_amd as dp
from ddsp_piano_amd import core
from util import synth_controls, synth_ir
rng = np.random.default_rng(0)
def group(P, sr):
    a = dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True)
    n = dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr)
    r = dp.Reverb(name='reverb')
    return dp.ProcessorGroup(dp.polyphonic_dag(a, n, r, additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'],
        noise_controls=['magnitudes'], reverb_controls=['reverb_ir'], n_synths=P))
def feats(B, P, T, H, K, L):
    f = {}
    for i in range(P):
        for k, v in synth_controls(rng, B, T, H, S=1, K=K).items():
            f[f'{k}_{i}'] = torch.as_tensor(v, device='cuda')
    f['reverb_ir'] = torch.as_tensor(synth_ir(rng, B, L), device='cuda')
    return f
for name, args in [('T=1', (2, 2, 1, 64, 32, 100)), ('T=2', (1, 1, 2, 64, 32, 7)), ('T=3,P=1,L=1', (1, 1, 3, 16, 32, 1)), ('H=1', (1, 2, 5, 1, 32, 50)),
                   ('H=200', (1, 2, 5, 200, 64, 50)), ('P=17', (1, 17, 4, 32, 32, 50)), ('B=0', (0, 2, 5, 32, 32, 50))]:
    try:
        B, P, T, H, K, L = args
        y = group(P, 24000)(feats(*args))
        torch.cuda.synchronize()
        print(name, 'ok', tuple(y.shape), bool(torch.isfinite(y).all()))
    except Exception as e:
        print(name, type(e).__name__, str(e)[:150])
