#!/usr/bin/env python3
"""the HIP FDN impulse response against the reference-faithful complex64 oracle, as a DISTRIBUTION over
networks drawn from the reference's initialisers (sub_modules.py:386-418) instead of a bound on six of them.

For each of N networks: impulse response by the library (float64 solve = the default, and the complex64-inverse switch), by
the oracle with exact_solve=False (tf.linalg.inv in complex64 as fdn_reverb.py:314-333 writes it) and with the float64 solve;
the AUDIO after ddsp.effects.Reverb on a piano-like dry signal; relative RMS errors.  Reported, not asserted.
usage: python tools/fdn_error_distribution.py [N=64] [sr=16000] > profiles/r04_fdn_error_distribution.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ddsp_piano_amd as dp  # noqa: E402
from ddsp_piano_amd import core  # noqa: E402
from util import O, rms, rms_err  # noqa: E402
from test_gpu_reverb_models import _tables  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sr = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
rng = np.random.default_rng(2024)
tables = _tables(rng, n)                       # T60 ~ N(2.0, 0.5) s, gains ~ N(0.25, 0.1): the initialiser's "lively" rooms
pm = np.arange(n, dtype=np.int32)[:, None]
dry = (rng.normal(0, 0.1, [n, 2 * sr]) * np.exp(-np.arange(2 * sr) / 4000.0)[None]).astype(np.float32)
layer = dp.MultiInstrumentFeedbackDelayReverb(n_instruments=n, sample_rate=sr)
layer.load_parameters(tables)
ref = {False: O.MultiInstrumentFeedbackDelayReverb(tables, n, sr, exact_solve=False)(pm),
       True: O.MultiInstrumentFeedbackDelayReverb(tables, n, sr, exact_solve=True)(pm)}
ref_audio = {k: O.Reverb().get_signal(dry, v) for k, v in ref.items()}
rows = {}
for mode in ('float64', 'complex64'):
    prev = core.set_recalled(fdn_solve=mode)
    try:
        ir = layer.call(torch.as_tensor(pm, device='cuda'))
    finally:
        core.set_recalled(**prev)
    audio = dp.Reverb().get_signal(torch.as_tensor(dry, device='cuda'), ir).cpu().numpy()
    irn = ir.cpu().numpy()
    for name, exact in (('complex64 oracle', False), ('float64 oracle', True)):
        rows[(mode, name, 'impulse response')] = np.asarray([rms_err(irn[b], ref[exact][b]) / rms(ref[exact][b]) for b in range(n)])
        rows[(mode, name, 'audio')] = np.asarray([rms_err(audio[b], ref_audio[exact][b]) / rms(ref_audio[exact][b]) for b in range(n)])
rows[('oracle float64', 'complex64 oracle', 'audio')] = np.asarray(
    [rms_err(ref_audio[True][b], ref_audio[False][b]) / rms(ref_audio[False][b]) for b in range(n)])
print(f'# FDN impulse responses of {n} networks drawn from the reference\'s initialisers (sub_modules.py:386-418), {sr} Hz, 2 s;')
print('# relative RMS error of the library (kernel solve mode) against the oracle, per quantity; quantiles over the networks')
print(f'# T60 of the draw: min {tables["time_rev_0_sec"].min():.2f} s, median {np.median(tables["time_rev_0_sec"]):.2f} s, '
      f'max {tables["time_rev_0_sec"].max():.2f} s')
print(f'{"kernel solve":<16}{"against":<20}{"quantity":<18}{"min":>10}{"median":>10}{"90 %":>10}{"99 %":>10}{"max":>10}   > 1e-4')
for (mode, name, what), e in rows.items():
    q = np.quantile(e, [0.0, 0.5, 0.9, 0.99, 1.0])
    print(f'{mode:<16}{name:<20}{what:<18}' + ''.join(f'{x:10.2e}' for x in q) + f'   {int((e > 1e-4).sum())} of {n}')
