"""GPU: the C-ABI's threading contract ("re-entrant per stream, no global mutable state except a plan cache
guarded by a mutex"; include/ddspp.h).  The reference's callers are one Python thread (synthesize_midi_file.py:73); a
consumer of libddspp.so need not be: here two host threads, each with its own HIP stream and its own ddspp_group handle,
push interleaved ddspp_group_run calls at BASELINE config 2's size (one pair shares every plan-cache key: same dims, same
reverb transform sizes), a third thread flips a tuning option meanwhile, and every result must equal the serial run's
bit for bit.  ctypes releases the GIL for the duration of a foreign call, so the calls really overlap inside the library.
"""
import ctypes
import threading

import numpy as np
import pytest
import torch

from test_gpu_native_group import _setup

pytestmark = pytest.mark.gpu


def _serial(dp, group, feats, z, calls):
    nat = dp.NativeGroup(group(), feats)
    out = [nat(feats, noise=z).clone() for _ in range(calls)]
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize('same_shapes', [True, False])
def test_two_threads_two_streams_two_groups(same_shapes):
    calls = 50
    # config 2: one 3 s segment, poly 16, 96 harmonics (and the H = 128 / K = 96 variant), 3 s reverb IR
    shapes = [(11, 1, 16, 750, 96, 64, 1, 96, False, 72000), (12, 1, 16, 750, 128, 96, 1, 96, False, 72000)]
    if same_shapes:
        shapes = [shapes[0], (13,) + shapes[0][1:]]           # same dims, other data: every plan-cache key is shared
    jobs = []
    for shp in shapes:
        dp, group, feats, _, noise, sr = _setup(*shp)
        z = torch.as_tensor(noise, device='cuda')
        jobs.append((dp, group, feats, z))
    want = [_serial(dp, group, feats, z, 2) for dp, group, feats, z in jobs]
    for w in want:                                            # (the run is deterministic call to call: noise is supplied)
        assert torch.equal(w[0], w[1])
    torch.cuda.synchronize()

    results = [[None] * calls for _ in jobs]
    errors = []
    start = threading.Barrier(len(jobs) + 1)
    stop = threading.Event()

    def worker(i):
        try:
            dp, group, feats, z = jobs[i]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                nat = dp.NativeGroup(group(), feats)          # its own ddspp_group handle and workspace
                start.wait()
                for c in range(calls):
                    results[i][c] = nat(feats, noise=z).clone()
                stream.synchronize()
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))
            try:
                start.abort()
            except Exception:  # noqa: BLE001
                pass

    def option_flipper():
        # an unrelated option set / read while the others launch: the option table is the library's other piece of shared
        # state (lock-free on the launch path since round 6, csrc/error.cpp)
        from ddsp_piano_amd import _lib
        lib = _lib.load()
        start.wait()
        k = 0
        while not stop.is_set():
            lib.ddspp_set_option(b'DDSPP_TEST_UNUSED_OPTION', k)
            assert lib.ddspp_option(b'DDSPP_TEST_UNUSED_OPTION', -1) == k
            k += 1

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(jobs))]
    flip = threading.Thread(target=option_flipper)
    for t in threads:
        t.start()
    flip.start()
    for t in threads:
        t.join()
    stop.set()
    flip.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for i in range(len(jobs)):
        for c in range(calls):
            assert torch.equal(results[i][c], want[i][0]), (i, c)


def test_last_error_is_thread_local():
    """include/ddspp.h: ddspp_last_error() is the message of the last failed call OF THE CALLING THREAD."""
    from ddsp_piano_amd import _lib
    lib = _lib.load()
    seen = {}
    go = threading.Barrier(2)

    def bad_call(tag, rows):
        go.wait()
        for _ in range(200):
            rc = lib.ddspp_cos_oscillator_bank(None, None, None, rows, 8, 64, ctypes.c_float(16000.0), 1, 1, 0, None, 0, None) \
                if tag == 'null' else \
                lib.ddspp_cos_oscillator_bank(ctypes.c_void_p(256), ctypes.c_void_p(256), ctypes.c_void_p(256), rows, 7, 64,
                                              ctypes.c_float(16000.0), 1, 1, 0, None, 0, None)
            assert rc == _lib.DDSPP_EINVAL
            msg = lib.ddspp_last_error().decode()
            seen.setdefault(tag, set()).add(msg)

    a = threading.Thread(target=bad_call, args=('null', 1))
    b = threading.Thread(target=bad_call, args=('dims', 2))
    a.start(); b.start(); a.join(); b.join()
    assert len(seen['null']) == 1 and 'null buffer' in next(iter(seen['null'])), seen
    assert len(seen['dims']) == 1 and 'multiple of 8' in next(iter(seen['dims'])), seen
