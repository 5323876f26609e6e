"""Shared helpers for the parity tests (test infrastructure; may import the oracle)."""
import numpy as np

from oracle import ddsp_oracle as O


def rms(x):
    x = np.asarray(x, np.float64)
    return float(np.sqrt(np.mean(x * x)))


def rms_err(a, b):
    return rms(np.asarray(a, np.float64) - np.asarray(b, np.float64))


def synth_controls(rng, B, T, H, S=1, K=None, silent_frac=0.25, midi_lo=21, midi_hi=108):
    """Synthetic post-network controls of one voice, SURVEY.md 8(d) recipe (raw, pre scale_fn)."""
    m = rng.integers(midi_lo, midi_hi + 1, size=[B, 1, 1]).astype(np.float64)
    f0 = 440.0 * 2.0 ** ((m - 69.0) / 12.0)
    detune = 2.0 ** (0.3 * np.arange(S)[None, None, :] / 1200.0)
    f0 = np.broadcast_to(f0 * detune, [B, T, S]).copy()
    silent = rng.random([B, 1, 1]) < silent_frac
    f0 = np.where(silent, 0.0, f0)
    inharm = np.exp(-0.105 * m - 6.87) + np.exp(0.094 * m - 13.70)
    inharm = np.broadcast_to(inharm, [B, T, 1]).copy()
    decay = np.exp(-np.arange(T)[None, :, None] / (0.4 * T))
    amps = rng.normal(-1.0, 1.0, [B, 1, 1]) * np.ones([1, T, 1]) + 3.0 * (decay - 1.0)
    hd = rng.normal(0.0, 1.0, [B, T, H]) - 0.05 * np.arange(1, H + 1)[None, None, :]
    kernel = np.ones(5) / 5.0
    hd = np.apply_along_axis(lambda v: np.convolve(np.pad(v, 2, mode='edge'), kernel, 'valid'), 1, hd)
    out = dict(amplitudes=amps.astype(np.float32), harmonic_distribution=hd.astype(np.float32),
               inharm_coef=inharm.astype(np.float32), f0_hz=f0.astype(np.float32))
    if K:
        out['magnitudes'] = rng.normal(0.0, 1.0, [B, T, K]).astype(np.float32)
    return out


def synth_ir(rng, B, L):
    n = np.arange(L)
    ir = rng.normal(0.0, 1.0, [B, L]) * np.exp(-6.9 * n / L)[None, :] * 0.02
    ir[:, 1] = 3.0
    return ir.astype(np.float32)


def set_option(monkeypatch, name, value=None):
    """Flip a DDSPP_* tuning / route switch for the rest of the test: the environment variable is changed AND the
    package is told to re-read its options (neither the library nor the host layer reads the environment per call)."""
    if value is None:
        monkeypatch.delenv(name, raising=False)
    else:
        monkeypatch.setenv(name, str(value))
    from ddsp_piano_amd import _lib
    _lib.options.reload()


__all__ = ['O', 'rms', 'rms_err', 'synth_controls', 'synth_ir', 'set_option']


def oracle_segments(feats, noises, P, sr, segments, threads=None, **flags):
    """The oracle's polyphonic group on the chosen batch rows of numpy features, one thread per voice task
    (numpy releases the GIL in its inner loops; a 3 s voice is ~1.5 s of single-core work).

    feats: {key_i: [B, T, C], 'reverb_ir': [B, L]}; noises: [B, P, N].  Returns {b: dict(signal, dry, additive_last,
    noise_last)} with the add chain in the DAG's order ((add + noise_i) + additive_i), polyphonic_dag.py:28-37."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    additive = O.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True, **flags)
    noise = O.FilteredNoise(name='noise', frame_rate=250, sample_rate=sr,
                            **({'scale_fn': flags['scale_fn']} if 'scale_fn' in flags else {}))
    reverb = O.Reverb()

    def voice(task):
        b, i = task
        sl = slice(b, b + 1)
        a = additive(feats[f'amplitudes_{i}'][sl], feats[f'harmonic_distribution_{i}'][sl],
                     feats[f'inharm_coef_{i}'][sl], feats[f'f0_hz_{i}'][sl])
        z = noise.get_signal(**noise.get_controls(feats[f'magnitudes_{i}'][sl]), noise=noises[sl, i])
        return a, z

    tasks = [(b, i) for b in segments for i in range(P)]
    with ThreadPoolExecutor(max_workers=threads or min(32, os.cpu_count() or 1)) as ex:
        sigs = list(ex.map(voice, tasks))
    out = {}
    for si, b in enumerate(segments):
        mix = None
        for a, z in sigs[si * P:(si + 1) * P]:
            mix = (z + a).astype(np.float32) if mix is None else ((mix + z).astype(np.float32) + a).astype(np.float32)
        a, z = sigs[si * P + P - 1]
        wet = reverb.get_signal(mix, feats['reverb_ir'][b:b + 1]) if 'reverb_ir' in feats else None
        out[b] = dict(signal=wet, dry=mix, additive_last=a, noise_last=z)
    return out


__all__ += ['oracle_segments']
