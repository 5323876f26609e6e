"""GPU: a caller that has ONLY include/ddspp.h and libddspp.so (tests/cabi/standalone.cpp: C++, HIP runtime, no Python,
no torch, tables from the library's own host builders) renders the committed golden cases; its audio is compared
with the golden audio here.  This is the binding INTEGRATION.md section 2 describes, exercised end to end."""
import os
import subprocess

import numpy as np
import pytest

from util import rms, rms_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
BIN = os.path.join(ROOT, 'ddsp_piano_amd', 'build', 'cabi_standalone')
TOL = 1e-4


def _binary():
    if not os.path.exists(BIN):
        import __graft_entry__ as entry          # builds the library and this program (hipcc)
        entry.build_cabi_standalone()
    assert os.path.exists(BIN), 'tests/cabi/standalone.cpp was not built (run __graft_entry__.build())'
    return BIN


def _run(case, d):
    p = subprocess.run([_binary(), case, str(d)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert 'gfx950' in p.stdout


def test_config1_through_the_header_alone(tmp_path):
    g = np.load(os.path.join(GOLD, 'c1_mono.npz'))
    T, H = g['raw_harmonic_distribution'].shape[1:]
    sr = int(g['sample_rate'])
    for k in ('raw_amplitudes', 'raw_harmonic_distribution', 'raw_inharm_coef', 'raw_f0_hz'):
        g[k].astype('<f4').tofile(tmp_path / f'{k}.f32')
    (tmp_path / 'dims.txt').write_text(f'{T}\n{H}\n{sr}\n{sr // int(g["frame_rate"])}\n')
    _run('c1', tmp_path)
    audio = np.fromfile(tmp_path / 'audio.f32', '<f4')[None, :]
    assert audio.shape == g['audio'].shape and rms_err(audio, g['audio']) < TOL


def test_small_config2_full_chain_through_the_header_alone(tmp_path):
    g = np.load(os.path.join(GOLD, 'c2_small.npz'))
    P, sr = int(g['n_synths']), int(g['sample_rate'])
    T, H = g['in_harmonic_distribution_0'].shape[1:]
    K, S, L = g['in_magnitudes_0'].shape[2], g['in_f0_hz_0'].shape[2], g['in_reverb_ir'].shape[1]
    for i in range(P):
        for k in ('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz', 'magnitudes'):
            g[f'in_{k}_{i}'].astype('<f4').tofile(tmp_path / f'{k}_{i}.f32')
        g['noises'][i].astype('<f4').tofile(tmp_path / f'noise_{i}.f32')
    g['in_reverb_ir'].astype('<f4').tofile(tmp_path / 'reverb_ir.f32')
    (tmp_path / 'dims.txt').write_text('\n'.join(str(v) for v in (P, T, H, K, S, sr, sr // int(g['frame_rate']), L)) + '\n')
    _run('c2', tmp_path)
    audio = np.fromfile(tmp_path / 'audio.f32', '<f4')[None, :]
    dry = np.fromfile(tmp_path / 'dry.f32', '<f4')[None, :]
    assert rms_err(dry, g['dry']) < TOL
    assert rms_err(audio, g['audio']) < TOL * max(1.0, rms(g['audio']))
    # the same segment through the one-call driver (ddspp_group_create / ddspp_group_run), outputs dictionary included
    audio_g = np.fromfile(tmp_path / 'audio_group.f32', '<f4')[None, :]
    dry_g = np.fromfile(tmp_path / 'dry_group.f32', '<f4')[None, :]
    assert rms_err(dry_g, g['dry']) < TOL
    assert rms_err(audio_g, g['audio']) < TOL * max(1.0, rms(g['audio']))
    assert rms_err(audio_g, audio) < 1e-6 * max(1.0, rms(audio))
    last = np.fromfile(tmp_path / 'additive_last_group.f32', '<f4') + np.fromfile(tmp_path / 'noise_last_group.f32', '<f4')
    assert np.isfinite(last).all() and rms(last) > 0
