import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
import numpy as np, torch
import ddsp_piano_amd as dp
from ddsp_piano_amd import core
from util import synth_controls
rng = np.random.default_rng(77)
sr, B, P, T, H, S = 24000, 16, 16, 375, 128, 1
U = 96
R = B * P
tt = np.arange(T)
pitch = np.where(tt < 125, 1000.0 * (1.0 + 0.1 * tt / 125.0), np.where(tt < 250, 1100.0, 55.0))
f0 = torch.as_tensor((pitch[None, :, None] * (1 + 0.01 * rng.random([R, 1, 1]))).astype(np.float32), device='cuda')
inh = torch.full([R, T, 1], 1e-4, device='cuda')
amp = torch.zeros([R, T, 1], device='cuda'); hd = torch.randn([R, T, H], device='cuda')
add = dp.MultiInharmonic(frame_rate=250, sample_rate=sr, inference=True)
def render(lo, hi, state=None, use_aud=True, spans=0):
    sl = slice(lo, hi)
    ctl = add._controls(amp[:, sl].contiguous(), hd[:, sl].contiguous(), inh[:, sl].contiguous(), f0[:, sl].contiguous(), want_counts=True, want_shifts=False)
    Tc = hi - lo
    return core.polyphonic_additive(ctl['f0_hz'], ctl['amplitudes'].reshape(R, Tc), ctl['harmonic_distribution'], None, B, Tc * U, sr,
                                    audible=ctl['_audible'] if use_aud else None, inharm_coef=ctl['_inharm_coef'].reshape(R, Tc), phase_state=state, spans=spans), ctl
whole, _ = render(0, T)
def st(lo, hi, nch, state=None, audible=None):
    return core.oscillator_phase_state(f0[:, lo:hi].contiguous(), nch, U, sr, inharm_coef=inh[:, lo:hi, 0].contiguous(), n_harmonics=H, phase_state=state, audible=audible)
s12 = st(0, 126, 12)
for use_aud in (True, False):
    for spans in (0, 1, 4):
        got, _ = render(125, 251, state=s12, use_aud=use_aud, spans=spans)
        d = (got[:, :12000] - whole[:, 12000:24000]).abs()
        print('aud', use_aud, 'spans', spans, 'err frames 0..123', d[:, :124 * U].max().item(), 'frame 124', d[:, 124 * U:].max().item())
