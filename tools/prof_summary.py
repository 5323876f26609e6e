#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table (text).
usage: tools/prof_summary.py results.db [> profiles/xxx.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('void ', '')
    return name[:110]


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, (end - start) from kernels").fetchall()
    agg = {}
    for name, d in rows:
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1.0
    print(f'{"kernel":112s} {"calls":>6s} {"total_ms":>10s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s}')
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{short(name):112s} {a[0]:6d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.1f} {a[2] / 1e3:10.1f} '
              f'{a[3] / 1e3:10.1f} {100 * a[1] / total:6.2f}')


if __name__ == '__main__':
    main(sys.argv[1])
