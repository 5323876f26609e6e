#!/usr/bin/env python3
"""profiles/osc_traffic.json from the two HBM counter passes over the graded kernel (separate rocprofv3 --pmc FETCH_SIZE /
--pmc WRITE_SIZE runs of tools/bench_kernels.py --which osc --spans 1): per-launch HBM bytes with the gfx950 correction
(FETCH_SIZE counts 64 B per 128 B request -> x2, MI355X_MICROARCH.md) and the hash of the kernel sources they describe.
usage: osc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [rows N H]"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddsp_piano_amd import _lib  # noqa: E402


def avg(path, counter, filt='osc_stream_kernel<2'):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if filt in row['Kernel_Name'] and row['Counter_Name'] == counter:
                vals.append(float(row['Counter_Value']))
    return sum(vals) / len(vals)


def main(fetch_csv, write_csv, out, rows=1024, n=72000, h=128):
    fetch, write = avg(fetch_csv, 'FETCH_SIZE'), avg(write_csv, 'WRITE_SIZE')
    hbm = 2.0 * fetch * 1024 + write * 1024
    alg = rows * (n * h * 8 + n * 4)
    json.dump({'csrc_hash': _lib.source_hash(), 'hbm_bytes_per_launch': hbm, 'fetch_size_kib': fetch, 'write_size_kib': write,
               'algorithmic_bytes_per_launch': alg, 'traffic_over_algorithmic': hbm / alg,
               'note': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on ddspp::osc_stream_kernel<2> '
                       f'at rows={rows}, N={n}, H={h}; FETCH_SIZE counts 64 B per 128 B request on gfx950 -> x2 '
                       '(MI355X_MICROARCH.md)'}, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3], *[int(a) for a in sys.argv[4:7]])
