#!/usr/bin/env python3
"""Per-kernel timing harness (HIP events) for the kernels of the synthesis path at bench sizes.
usage: python tools/bench_kernels.py [--which fused,compact,osc,fir,controls] [--batch 64] [--reps 5]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ddsp_piano_amd as dp  # noqa: E402
from ddsp_piano_amd import core  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--which', default='fused,osc,fir,controls')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--poly', type=int, default=16)
    ap.add_argument('--harmonics', type=int, default=128)
    ap.add_argument('--bands', type=int, default=96)
    ap.add_argument('--substrings', type=int, default=1)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--spans', default='0')
    ap.add_argument('--osc-rows', type=int, default=1024)
    args = ap.parse_args()
    which = set(args.which.split(','))
    dev = torch.device('cuda', 0)
    sr, U, T = 24000, 96, 750
    N = T * U
    B, P, H, K, S = args.batch, args.poly, args.harmonics, args.bands, args.substrings
    R = B * P
    feats, base = bench.make_features(B, P, T, H, K, S, 72000, dev, 1)
    additive = dp.MultiInharmonic(sample_rate=sr, inference=True)
    noise = dp.DynamicSizeFilteredNoise(sample_rate=sr)
    amp = base['amplitudes'].reshape(R, T, 1)
    hd = base['harmonic_distribution'].reshape(R, T, H)
    inh = base['inharm_coef'].reshape(R, T, 1)
    f0 = base['f0_hz'].reshape(R, T, S)
    mags = base['magnitudes'].reshape(R, T, K)
    osc_total = R * S * N * H

    if 'controls' in which:
        mn, av = timeit(lambda: additive._controls(amp, hd, inh, f0), args.reps)
        print(f'inharmonic_controls      min {mn:8.3f} ms avg {av:8.3f} ms')
        mn, av = timeit(lambda: noise.get_controls(mags), args.reps)
        print(f'noise get_controls       min {mn:8.3f} ms avg {av:8.3f} ms')
    ctl = additive._controls(amp, hd, inh, f0)
    if 'fused' in which:
        for sp in [int(s) for s in args.spans.split(',')]:
            out = torch.empty((R, N), device=dev)
            fn = lambda: core.harmonic_synthesis_fused(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T),  # noqa: E731
                                                       ctl['harmonic_distribution'], ctl['harmonic_shifts'], N,
                                                       sr, True, spans=sp, out=out)
            mn, av = timeit(fn, args.reps)
            print(f'harmonic_synthesis fused spans={sp:3d} min {mn:8.3f} ms avg {av:8.3f} ms  '
                  f'{osc_total / mn / 1e6:8.1f} G osc-samples/s')
    if 'compact' in which:
        for sp in [int(s) for s in args.spans.split(',')]:
            fn = lambda: core.polyphonic_additive(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T),  # noqa: E731
                                                  ctl['harmonic_distribution'], ctl['harmonic_shifts'], B, N, sr,
                                                  spans=sp)
            mn, av = timeit(fn, args.reps)
            print(f'polyphonic_additive spans={sp:3d} min {mn:8.3f} ms avg {av:8.3f} ms')
    if 'fir' in which:
        nctl = noise.get_controls(mags)['magnitudes']
        mn, av = timeit(lambda: core.frequency_impulse_response(nctl, 257), args.reps)
        print(f'fir_from_magnitudes      min {mn:8.3f} ms avg {av:8.3f} ms')
        ir = core.frequency_impulse_response(nctl, 257)
        x = core.uniform_noise((R, N), seed=1, device=dev)
        mn, av = timeit(lambda: core.fft_convolve(x, ir), args.reps)
        print(f'time_varying_fir         min {mn:8.3f} ms avg {av:8.3f} ms')
    if 'noise' in which:
        x = core.uniform_noise((R, N), seed=1, device=dev)
        rs = noise.raw_scale()
        mn, av = timeit(lambda: core.frequency_filter(x, mags, window_size=noise.window_size, raw_scale=rs), args.reps)
        print(f'frequency_filter (raw magnitudes -> filtered noise) min {mn:8.3f} ms avg {av:8.3f} ms')
    if 'noisev' in which:                 # the batched group's call: voice sums, last voice split off or not
        x = core.uniform_noise((R, N), seed=1, device=dev)
        rs = noise.raw_scale()
        for vq in (8, 4, 2):
            for split in (False, True):
                fn = lambda: core.frequency_filter_voice_sums(x, mags, noise.window_size, rs, P, vq, False, split_last=split)  # noqa: E731
                mn, av = timeit(fn, args.reps)
                print(f'frequency_filter_voice_sums vq={vq} split_last={int(split)} min {mn:8.3f} ms avg {av:8.3f} ms')
    if 'osc' in which:
        rows = min(args.osc_rows, R)
        c = additive._controls(amp[:rows], hd[:rows], inh[:rows], f0[:rows, :, :1].contiguous())
        hf = core.get_harmonic_frequencies(c['f0_hz'], H) * (1.0 + c['harmonic_shifts'])
        ha = c['amplitudes'] * c['harmonic_distribution']
        mn, av = timeit(lambda: core.resample(hf, N), 2)
        print(f'resample linear          min {mn:8.3f} ms  ({rows * N * H * 4 / mn / 1e6:7.1f} GB/s written)')
        mn, av = timeit(lambda: core.resample(ha, N, method='window'), 2)
        print(f'resample window          min {mn:8.3f} ms  ({rows * N * H * 4 / mn / 1e6:7.1f} GB/s written)')
        fe = core.resample(hf, N)
        ae = core.resample(ha, N, method='window')
        for sp in [int(s) for s in args.spans.split(',')]:
            mn, av = timeit(lambda: core.cos_oscillator_bank(fe, ae, sr, True, True, spans=sp or 1), args.reps)
            byts = rows * (N * H * 8 + N * 4)
            print(f'cos_oscillator_bank spans={sp or 1:3d} rows={rows} min {mn:8.3f} ms avg {av:8.3f} ms  '
                  f'{byts / mn / 1e6:7.1f} GB/s algorithmic = {byts / mn / 1e6 / 8000:.3f} of 8 TB/s')


if __name__ == '__main__':
    main()
