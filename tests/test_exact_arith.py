"""CPU: the exact-arithmetic shortcuts of the kernels (constant division, 2*pi reduction) are checked
against IEEE division / fmod on a dense sample of all float32 inputs (exhaustive run: stride 1)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc')
def test_exact_arith_checker():
    subprocess.run(['make', '-s', '-C', os.path.join(ROOT, 'oracle')], check=True)
    out = subprocess.run([os.path.join(ROOT, 'oracle', 'exact_arith_check'), '97'], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'mod_2pi fast path: 0 mismatches' in out.stdout


def test_markstein_division_numpy_spot_check():
    """Same identity, restated in numpy with float64 FMA emulation, on the frequencies a piano produces."""
    rng = np.random.default_rng(0)
    x = (rng.uniform(20.0, 2.0e6, 200000).astype(np.float32) * np.float32(2 * np.pi)).astype(np.float32)
    for d in (16000.0, 24000.0, 48000.0):
        d32 = np.float32(d)
        rd = np.float32(1.0) / d32
        q = (x * rd).astype(np.float32)
        r = (x.astype(np.float64) - q.astype(np.float64) * np.float64(d32)).astype(np.float32)   # exact FMA
        y = (q.astype(np.float64) + r.astype(np.float64) * np.float64(rd)).astype(np.float32)
        assert np.array_equal(y, (x / d32).astype(np.float32))
