import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np, torch
import ddsp_piano_amd as dp
from ddsp_piano_amd import core
rng = np.random.default_rng(1)
sr, R, T, H, S = 24000, 256, 375, 128, 1
U = 96
tt = np.arange(T)
pitch = np.where(tt < 125, 1000.0 * (1.0 + 0.1 * tt / 125.0), np.where(tt < 250, 1100.0, 55.0))
f0 = torch.as_tensor((pitch[None, :, None] * (1 + 0.01 * rng.random([R, 1, 1]))).astype(np.float32), device='cuda')
inh = torch.full([R, T], 1e-4, device='cuda')
def st(Tc, nch, rows=R, audible=None):
    return core.oscillator_phase_state(f0[:rows, :Tc].contiguous(), nch, U, sr, inharm_coef=inh[:rows, :Tc].contiguous(), n_harmonics=H, audible=audible)
a = st(126, 12)
b = st(375, 12)
c = st(126, 12, rows=8)
print('T126 vs T375', (a - b).abs().max().item(), 'memo vs chunk-parallel', (a[:8] - c).abs().max().item())
add = dp.MultiInharmonic(frame_rate=250, sample_rate=sr, inference=True)
amp = torch.zeros([R, T, 1], device='cuda'); hd = torch.randn([R, T, H], device='cuda')
ctl = add._controls(amp[:, :126].contiguous(), hd[:, :126].contiguous(), inh[:, :126, None].contiguous(), f0[:, :126].contiguous(), want_counts=True, want_shifts=False)
d = st(126, 12, audible=ctl['_audible'])
print('with audible', (a - d).abs().max().item(), (a-d).abs().max(0)[0].reshape(-1)[::8].tolist()[:16])
