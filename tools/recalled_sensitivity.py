#!/usr/bin/env python3
"""What each recalled ddsp detail is WORTH (VERDICT r04 next #2c): for every switch of oracle.ddsp_oracle.RECALLED, the RMS
difference of the rendered audio between its two settings, on

  C2      one 3 s poly-16 segment, 24 kHz, H=96, K=64, 3 s impulse response  (BASELINE config 2)
  C3-seg  one segment of config 3: H=128, K=96 (held notes, as bench.py's headline inputs: `resize` cannot matter)
  C3-vib  the same with every f0 moving in every frame (0.2 % vibrato at 5 Hz + a slow glide: bench.py's `moving_f0`)
  file    a 136 s poly-16 file in one segment (what synthesize_midi_file.py renders), H=128, K=96, 2 s impulse response,
          pitches stepping every 0.6 s with vibrato (tests/test_gpu_long_file.py's voices)

against the parity bar of BASELINE.json (1e-4 RMS).  A switch whose settings differ by less than the bar cannot break
parity whichever recollection is right; one that differs by more decides it, and the golden case that settles it on a TF
host is named next to it (tests/golden/make_golden.py prints which setting the real library matches).

CPU only (the numpy oracle; test infrastructure).  A branch is re-rendered only for the switches that reach it: the additive
branch for resize / angular_cumsum / angular_offsets / exp_sigmoid, the noise branch for auto_delay / window_crop /
exp_sigmoid / initial_bias.  `python tools/recalled_sensitivity.py [--cases C2,C3-seg,file] [--threads N] > profiles/recalled_sensitivity.txt`"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ddsp_oracle as O  # noqa: E402
from util import rms, synth_controls, synth_ir  # noqa: E402

CASES = {          # name: (T frames, H, K, IR samples)
    'C2': (750, 96, 64, 72000),
    'C3-seg': (750, 128, 96, 72000),
    'C3-vib': (750, 128, 96, 72000),
    'file': (34000, 128, 96, 48000),
}
P, SR = 16, 24000
# (switch, alternative setting, branches it reaches, the golden case that decides it on a TF host)
SWITCHES = [
    ('auto_delay', 'half', ('noise',), "recalled_details.npz: flat_full (a flat spectrum comes out delayed by 2 samples or by 0)"),
    ('window_crop', 'centred', ('noise',), "recalled_details.npz: flat_crop (only when 2 (K - 1) > window_size = 257: no shipped configuration)"),
    ('resize', 'half_pixel', ('additive',), "recalled_details.npz: ramp_linear, rs_linear_96, rs_linear_nonint (bitwise)"),
    ('angular_cumsum', 'exclusive', ('additive',), "recalled_details.npz: phase, phase_long_* (bitwise)"),
    ('angular_offsets', 'plain', ('additive',), "recalled_details.npz: phase_long_strided / phase_long_tail (bitwise, 301 chunks)"),
    ('angular_wrap', 'none', ('additive',), "recalled_details.npz: phase, phase_long_* (the function's output lies in [0, 2 pi) or grows to ~3000 rad: no tolerance needed)"),
    ('exp_sigmoid', (10.0, 2.0, 0.0), ('additive', 'noise'), "recalled_details.npz: exp_sigmoid (threshold 1e-7 -> 0 shown here)"),
    ('initial_bias', -4.0, ('noise',), "recalled_details.npz: noise_controls (-5 -> -4 shown here)"),
]


def render(case, threads, only=None, log=sys.stderr):
    T, H, K, L = CASES[case]
    N = T * (SR // 250)
    rng = np.random.default_rng(20240)
    voices = [synth_controls(rng, 1, T, H, S=1, K=K) for _ in range(P)]
    tt = np.arange(T, dtype=np.float64) / 250.0
    for v in voices:
        if case == 'C3-vib':
            v['f0_hz'] = (v['f0_hz'] * (1.0 + 0.002 * np.sin(2 * np.pi * 5.0 * tt + rng.uniform(0, 6.28)) + 0.0002 * tt)[None, :, None]).astype(np.float32)
        elif case == 'file':
            steps = 2.0 ** (rng.integers(-3, 4, size=[1, T // 150 + 1, 1]).repeat(150, axis=1)[:, :T] / 12.0)
            v['f0_hz'] = (v['f0_hz'] * steps * (1 + 0.002 * np.sin(np.arange(T) / 9.0))[None, :, None]).astype(np.float32)
            v['amplitudes'] = (v['amplitudes'] * 0 + rng.normal(-1.0, 0.3, [1, T, 1])).astype(np.float32)      # no decay over the file
    noises = [rng.uniform(-1, 1, [1, N]).astype(np.float32) for _ in range(P)]
    ir = synth_ir(rng, 1, L)

    def additive_stem(i):
        syn = O.MultiInharmonic(frame_rate=250, sample_rate=SR, inference=True)
        v = voices[i]
        return syn(v['amplitudes'], v['harmonic_distribution'], v['inharm_coef'], v['f0_hz'])

    def noise_stem(i):
        syn = O.FilteredNoise(frame_rate=250, sample_rate=SR)
        return syn.get_signal(**syn.get_controls(voices[i]['magnitudes']), noise=noises[i])

    def branch(fn, settings):
        # the oracle reads RECALLED at call time from a module global: one setting at a time, voices in parallel
        with O.recalled(**settings):
            with ThreadPoolExecutor(max_workers=threads) as ex:
                return list(ex.map(fn, range(P)))

    def mix(adds, nzs):
        m = None
        for a, z in zip(adds, nzs):            # polyphonic_dag.py:28-37: ((add + noise_i) + additive_i)
            m = (z + a).astype(np.float32) if m is None else ((m + z).astype(np.float32) + a).astype(np.float32)
        return m, O.Reverb().get_signal(m, ir)

    t0 = time.time()
    base_add, base_noise = branch(additive_stem, {}), branch(noise_stem, {})
    dry0, wet0 = mix(base_add, base_noise)
    print(f'[{case}] baseline rendered in {time.time() - t0:.0f} s', file=log, flush=True)
    rows = []
    for name, alt, reaches, decided_by in SWITCHES:
        if only and name not in only:
            continue
        t0 = time.time()
        adds = branch(additive_stem, {name: alt}) if 'additive' in reaches else base_add
        nzs = branch(noise_stem, {name: alt}) if 'noise' in reaches else base_noise
        with O.recalled(**{name: alt}):
            dry, wet = mix(adds, nzs)
        rows.append((name, O.RECALLED_DEFAULTS[name], alt, rms(dry - dry0), rms(wet - wet0), decided_by))
        print(f'[{case}] {name} = {alt!r}: {time.time() - t0:.0f} s', file=log, flush=True)
    return dict(case=case, T=T, H=H, K=K, L=L, N=N, rms_dry=rms(dry0), rms_wet=rms(wet0), rows=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='C2,C3-seg,C3-vib,file')
    ap.add_argument('--threads', type=int, default=max(1, min(8, os.cpu_count() or 1)))
    ap.add_argument('--file-threads', type=int, default=3, help='voices in flight for the 136 s file (each holds ~10 GB of [N, H] temporaries)')
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    only = [s for s in args.only.split(',') if s]
    bar = 1e-4
    print('# RMS audio difference between the two settings of every recalled ddsp detail (numpy oracle, float32-faithful).')
    print(f'# poly {P}, {SR} Hz, bench-style synthetic controls (seed 20240, tests/util.synth_controls); parity bar {bar:g} RMS.')
    print('# "breaks parity": the difference exceeds the bar, i.e. the wrong recollection would fail BASELINE.json\'s 1e-4.')
    for case in args.cases.split(','):
        r = render(case, args.file_threads if case == 'file' else args.threads, only)
        print(f'\n## {case}: {r["N"]} samples ({r["N"] / SR:g} s), H={r["H"]}, K={r["K"]}, IR {r["L"]} samples; '
              f'signal RMS dry {r["rms_dry"]:.3e}, wet {r["rms_wet"]:.3e}')
        print(f'{"detail":16s} {"default":22s} {"alternative":22s} {"rms diff (dry mix)":>19s} {"rms diff (output)":>18s} {"x bar":>8s}  breaks parity?  decided by')
        for name, dflt, alt, d_dry, d_wet, by in r['rows']:
            worst = max(d_dry, d_wet)
            print(f'{name:16s} {str(dflt):22s} {str(alt):22s} {d_dry:19.3e} {d_wet:18.3e} {worst / bar:8.2f}  '
                  f'{"YES" if worst > bar else "no":14s}  {by}')


if __name__ == '__main__':
    main()
