#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv. usage: pmc_summary.py file.csv [kernel-substring]"""
import csv
import re
import sys
from collections import defaultdict


def main(path, filt=''):
    agg = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = re.sub(r'\(.*$', '', row['Kernel_Name']).replace('void ', '')
            if filt and filt not in name:
                continue
            agg[name][row['Counter_Name']].append(float(row['Counter_Value']))
    for name, ctrs in agg.items():
        print(name[:100])
        for c, v in sorted(ctrs.items()):
            print(f'    {c:28s} n={len(v):3d} avg={sum(v) / len(v):16.1f} max={max(v):16.1f}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
