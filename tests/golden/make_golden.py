#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the BUILD container, where /root/reference exists).

    python tests/golden/make_golden.py

Writes (all small, float32 .npz):
  dafx22_reverb_ir.npz   rows 0 and 9 of the learned reverb bank [10, 24000] of the reference's dafx22
                         checkpoint (ddsp_piano/model_weights/dafx22/ckpt-0.data-00000-of-00001, raw
                         little-endian float32 at byte offset 308892, located by decoding ckpt-0.index;
                         SURVEY.md fact 6).  This is DATA shipped with the reference, not source.
  c1_mono.npz            BASELINE config 1: 1 s monophonic note, poly=1, 64 harmonics, 24 kHz, dry.
  c2_small.npz           down-sized config 2: B=1, P=2, T=50, H=96, K=64, S=2, 16 kHz, full chain with
                         the real dafx22 IR (first 6000 taps of row 0).
The expected outputs are "restatement goldens": they come from oracle/ddsp_oracle.py because
TensorFlow / ddsp cannot be imported here.  If a TF + ddsp host ever exists,
run this same script there with DDSP_GOLDEN_BACKEND=tf to upgrade them to TF goldens (hook below).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import ddsp_oracle as O  # noqa: E402
from util import synth_controls  # noqa: E402

REF_CKPT = '/root/reference/ddsp_piano/model_weights/dafx22/ckpt-0.data-00000-of-00001'
IR_OFFSET, IR_ROWS, IR_LEN = 308892, 10, 24000


def backend():
    if os.environ.get('DDSP_GOLDEN_BACKEND') == 'tf':
        raise SystemExit('TF backend hook: import ddsp / ddsp_piano here and build the same processors '
                         '(MultiInharmonic, DynamicSizeFilteredNoise, effects.Reverb) with the same inputs.')
    return O


def dafx22_ir():
    with open(REF_CKPT, 'rb') as f:
        f.seek(IR_OFFSET)
        bank = np.frombuffer(f.read(IR_ROWS * IR_LEN * 4), dtype='<f4').reshape(IR_ROWS, IR_LEN)
    assert abs(bank[0, 1] - 3.18) < 0.05, bank[0, :3]      # first taps ~ [-7.8e-5, 3.18, -1.3e-2]
    return np.ascontiguousarray(bank[[0, 9]])


def main():
    B_ = backend()
    ir = dafx22_ir()
    np.savez_compressed(os.path.join(HERE, 'dafx22_reverb_ir.npz'), ir=ir, rows=np.array([0, 9]))

    # ---- C1: 1 s mono note, 64 harmonics, 24 kHz, dry -----------------------------------------
    rng = np.random.default_rng(1234)
    T, H, sr = 250, 64, 24000
    raw = synth_controls(rng, 1, T, H, S=1, silent_frac=0.0, midi_lo=45, midi_hi=45)
    synth = B_.MultiInharmonic(frame_rate=250, sample_rate=sr, inference=True)
    ctl = synth.get_controls(raw['amplitudes'], raw['harmonic_distribution'], raw['inharm_coef'], raw['f0_hz'])
    audio = synth.get_signal(**ctl)
    np.savez_compressed(os.path.join(HERE, 'c1_mono.npz'), sample_rate=sr, frame_rate=250,
                        **{f'raw_{k}': v for k, v in raw.items()}, **{f'ctl_{k}': v for k, v in ctl.items()},
                        audio=audio.astype(np.float32))

    # ---- down-sized C2: B=1, P=2, T=50, dafx22 dims, full chain --------------------------------
    rng = np.random.default_rng(1234)
    B, P, T, H, K, S, sr = 1, 2, 50, 96, 64, 2, 16000
    N = T * (sr // 250)
    feats = {}
    for i in range(P):
        for k, v in synth_controls(rng, B, T, H, S=S, K=K, silent_frac=0.0).items():
            feats[f'{k}_{i}'] = v
    feats['reverb_ir'] = np.ascontiguousarray(ir[:1, :6000])
    noises = np.stack([rng.uniform(-1, 1, [B, N]).astype(np.float32) for _ in range(P)], axis=0)
    additive = B_.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True)
    noise = B_.FilteredNoise(name='noise', frame_rate=250, sample_rate=sr)
    dag = B_.polyphonic_dag(additive, noise, B_.Reverb(name='reverb'),
                            additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'],
                            noise_controls=['magnitudes'], reverb_controls=['reverb_ir'], n_synths=P)
    out = B_.ProcessorGroup(dag)(feats, return_outputs_dict=True,
                                 extra_kwargs={'noise': [{'noise': z} for z in noises]})
    np.savez_compressed(os.path.join(HERE, 'c2_small.npz'), sample_rate=sr, frame_rate=250, n_synths=P,
                        noises=noises, audio=out['signal'].astype(np.float32),
                        dry=out['controls']['add']['signal'].astype(np.float32),
                        **{f'in_{k}': v for k, v in feats.items()})
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')


if __name__ == '__main__':
    main()
