#!/usr/bin/env python3
"""Round 6: the two stand-alone upsamplers and the pure write ceiling beside them, config-3 dims (1024 rows x 72000 x 128).
usage: python tools/resample_time.py [rows]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ddsp_piano_amd import core  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device('cuda', 0)
    T, U, H = 750, 96, 128
    N = T * U
    g = torch.Generator(device=dev).manual_seed(1)
    hf = torch.rand((rows, T, H), device=dev, generator=g) * 8000.0
    ha = torch.rand((rows, T, H), device=dev, generator=g)
    nbytes = rows * N * H * 4
    lib = core._lib_()
    y = torch.empty((rows, N, H), device=dev)
    for nt in (1, 0):
        for waves in (2048, 4096, 8192, 16384):
            nw = ctypes.c_size_t(0)

            def wr():
                rc = lib.ddspp_hbm_write_probe(core._ptr(y), y.numel(), waves, nt, ctypes.byref(nw), core._stream())
                assert rc == 0
            wr()
            t = float(np.min(bench.event_times(wr, 4, warmup=1))) * 1e-3
            print(f'write probe nt={nt} streams={waves:6d}: {t * 1e3:7.3f} ms  {nw.value / t / 1e9:7.1f} GB/s')
    del y
    from ddsp_piano_amd import _lib
    for rep in range(2):
        for nt in (1, 0):
            _lib.set_option('DDSPP_RESAMPLE_NT', nt)
            for name, fn in (('resample linear', lambda: core.resample(hf, N)), ('resample window', lambda: core.resample(ha, N, method='window'))):
                ts = np.array(bench.event_times(fn, 5, warmup=1))
                print(f'{name} ({"non-temporal" if nt else "plain"} stores): min {ts.min():7.3f} ms  mean {ts.mean():7.3f} ms  '
                      f'{nbytes / ts.mean() / 1e6:7.1f} GB/s written')


if __name__ == '__main__':
    main()
