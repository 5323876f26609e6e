#!/usr/bin/env python3
"""Randomised cross-checks of the fast routes against the plain ones (bitwise where the arithmetic is the same):
   fused FilteredNoise kernel vs FIR design + time-varying FIR; compacted additive mix vs the sum of the stems;
   batched group vs node-by-node walk.  usage: python tools/fuzz_routes.py [n_cases]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ddsp_piano_amd as dp
from ddsp_piano_amd import _lib, core
from util import synth_controls, synth_ir
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2024)
bad = 0
for case in range(n_cases):
    U = int(rng.choice([96, 128, 192, 64])); sr = 250 * U
    K = int(rng.choice([32, 64, 96])); H = int(rng.choice([16, 64, 96, 128])); S = int(rng.choice([1, 1, 2]))
    B, P, T = int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.integers(2, 120))
    if os.environ.get('FUZZ_LONG'):
        T = int(rng.integers(400, 4000))                  # many chunks: pre-pass / scan variants
    N = T * U
    # 1. noise: fused vs split
    raw = torch.as_tensor(rng.normal(0, 2, [B * P, T, K]).astype(np.float32), device='cuda')
    x = torch.as_tensor(rng.uniform(-1, 1, [B * P, N]).astype(np.float32), device='cuda')
    syn = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=sr)
    a = core.frequency_filter(x, raw, window_size=syn.window_size, raw_scale=syn.raw_scale())
    os.environ['DDSPP_FIR_NO_FUSED'] = '1'
    _lib.options.reload()
    b = core.frequency_filter(x, raw, window_size=syn.window_size, raw_scale=syn.raw_scale())
    del os.environ['DDSPP_FIR_NO_FUSED']
    _lib.options.reload()
    ok1 = torch.equal(a, b)
    # 2. additive: compact mix vs stems
    ctlraw = synth_controls(rng, B * P, T, H, S=S, silent_frac=0.3)
    add = dp.MultiInharmonic(sample_rate=sr, inference=True)
    rt = [torch.as_tensor(ctlraw[k], device='cuda') for k in ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz')]
    ctl = add._controls(*rt, want_counts=True)
    amp = ctl['amplitudes'].reshape(B * P, T).contiguous()
    ok2 = True
    if core.fused_synthesis_supported(T, N) and P * S <= 64:
        stems = core.harmonic_synthesis_fused(ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'], N, sr, True).reshape(B, P, N)
        mix = core.polyphonic_additive(ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, sr, audible=ctl['_audible'])
        mix2 = core.polyphonic_additive(ctl['f0_hz'], amp, ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, sr)
        ok2 = (mix - stems.sum(1)).abs().max().item() < 5e-6 and torch.equal(mix, mix2)
    if not (ok1 and ok2):
        bad += 1
        print('MISMATCH', dict(U=U, K=K, H=H, S=S, B=B, P=P, T=T), ok1, ok2)
torch.cuda.synchronize()
print(f'{n_cases} cases, {bad} mismatches')
