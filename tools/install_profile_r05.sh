#!/bin/bash
# Copy what tools/profile_r05.sh left under gpurun_out/prof_r05 (merged back by gpurun) into profiles/ under the round's names,
# rebuild the reverb excerpt and the result blocks of DESIGN.md / BASELINE.md.  usage: bash tools/install_profile_r05.sh
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r05
cp $S/bench.json profiles/r05_bench.json
cp $S/bench_under_rocprof.json profiles/r05_bench_under_rocprof.json
cp $S/bench_kernel_stats.csv profiles/r05_bench_kernel_stats.csv
cp $S/osc_pmc.txt profiles/r05_osc_pmc.txt
cp $S/step_pmc_summary.txt profiles/r05_step_pmc.txt
cp $S/noise_pmc.txt profiles/r05_noise_pmc.txt
cp $S/step_valu.json $S/osc_traffic.json profiles/
python - <<'PY'
import re
hdr = [l for l in open('profiles/r05_reverb_pmc.txt') if l.startswith('#') and not l.startswith('# sum:')]
body, keep, ms, rd, wr = [], False, 0.0, 0.0, 0.0
for l in open('profiles/r05_step_pmc.txt'):
    if not l.startswith((' ', '\t', '#')):
        keep = l.startswith('part_')
        if keep:
            body.append(l)
        continue
    if not keep:
        continue
    t = l.strip()
    if t.startswith(('launches', 'FETCH_SIZE', 'WRITE_SIZE', '-> HBM')):
        body.append(l)
    m = re.match(r'launches\s+\d+\s+avg\s+([\d.]+) ms', t)
    if m:
        ms += float(m.group(1))
    m = re.match(r'-> HBM read ([\d.]+) GB, written ([\d.]+) GB', t)
    if m:
        rd, wr = rd + float(m.group(1)), wr + float(m.group(2))
alg = (72000 + 72000 + 72000) * 4 * 64 / 1e9
body.append(f'# sum: {ms:.3f} ms under the counters\' serialisation, HBM read {rd:.3f} GB + written {wr:.3f} GB = {rd + wr:.3f} GB = '
            f'{(rd + wr) / alg:.1f} x the algorithmic bytes (round 4: 0.131 ms, 0.322 GB, 5.8 x; the floor of this structure is 5.1 x)\n')
open('profiles/r05_reverb_pmc.txt', 'w').write(''.join(hdr + body))
PY
python tools/results_r05.py
python - <<'PY'
import json
d = json.loads(open('profiles/r05_bench.json').read().strip().splitlines()[-1])
print('source hash', d['roofline'].get('source_hash'), 'stale', d['roofline'].get('counters_stale'), 'ms_per_step', d['ms_per_step'])
PY
