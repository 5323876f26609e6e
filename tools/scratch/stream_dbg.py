import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
import numpy as np, torch
import test_gpu_streaming as t
import ddsp_piano_amd as dp
from ddsp_piano_amd import streaming
from util import synth_controls
KEYS = t.KEYS
rng = np.random.default_rng(77)
sr, B, P, T, H, K, S = 24000, 16, 16, 375, 128, 96, 1
U = sr // 250
feats = {}
tt = np.arange(T)
pitch = np.where(tt < 125, 1000.0 * (1.0 + 0.1 * tt / 125.0), np.where(tt < 250, 1100.0, 55.0))
for i in range(P):
    c = synth_controls(rng, B, T, H, S=S, K=K, silent_frac=0.0)
    c['f0_hz'] = (pitch[None, :, None] * (1.0 + 0.01 * rng.random([B, 1, 1]))).astype(np.float32) * np.ones([1, 1, S], np.float32)
    c['inharm_coef'] = np.full([B, T, 1], 1e-4, np.float32)
    c['amplitudes'] = np.zeros([B, T, 1], np.float32)
    for k, v in c.items():
        feats[f'{k}_{i}'] = torch.as_tensor(v, device='cuda')
noise = torch.zeros([B, P, T * U], device='cuda')
a, z, _ = t._processors(dp, sr)
whole = dp.ProcessorGroup(dp.polyphonic_dag(a, z, None, n_synths=P, **{**KEYS, 'reverb_controls': []}))(feats, noise=noise)
for cuts in ((126, 251, T), (251, T), (126, T)):
    syn = streaming.StreamingSynthesizer(*t._processors(dp, sr)[:2], None, n_synths=P)
    outs, t0 = [], 0
    for t1 in cuts:
        outs.append(syn.push({k: v[:, t0:t1] for k, v in feats.items()}, noise=noise[:, :, t0 * U:t1 * U], final=(t1 == T)))
        t0 = t1
    got = torch.cat(outs, dim=1)
    d = (got - whole).abs()
    print(cuts, [round(d[:, a0 * U:a1 * U].max().item(), 6) for a0, a1 in ((0, 125), (125, 249), (249, 250), (250, 260), (260, 375))], 'scale', whole.abs().max().item())
from util import oracle_segments
fn = {k: v.cpu().numpy() for k, v in feats.items()}
o = oracle_segments(fn, noise.cpu().numpy(), P, sr, [0])[0]['dry'][0]
w0 = whole[0].cpu().numpy()
g0 = got[0].cpu().numpy()
for nm, x in (('whole', w0), ('pieces(126,375)', g0)):
    d = np.abs(x - o)
    print(nm, 'vs oracle', [round(float(d[a0 * U:a1 * U].max()), 6) for a0, a1 in ((0, 125), (125, 249), (249, 250), (250, 260), (260, 375))])
