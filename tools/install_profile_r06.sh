#!/bin/bash
# Copy what tools/profile_r06.sh left under gpurun_out/prof_r06 (merged back by gpurun) into profiles/ under the round's names.
# usage: bash tools/install_profile_r06.sh
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r06
cp $S/bench.json profiles/r06_bench.json
cp $S/bench_under_rocprof.json profiles/r06_bench_under_rocprof.json
cp $S/bench_kernel_stats.csv profiles/r06_bench_kernel_stats.csv
cp $S/osc_pmc.txt profiles/r06_osc_pmc.txt
cp $S/step_pmc_summary.txt profiles/r06_step_pmc.txt
cp $S/step_valu.json $S/osc_traffic.json profiles/
{
  echo "# tools/ubench probes of round 6 (tools/profile_r06.sh), one MI355X"
  for f in osc_graded_spread osc_graded lds_stream store_cost resample_time philox_rates; do
    [ -f $S/$f.txt ] && { echo; echo "## $f"; cat $S/$f.txt; }
  done
  echo; echo "## side stream A/B (DDSPP_SIDE_STREAM=1; bench.py --steps 40 --sustain-seconds 3: synchronised median, min, sustained, pipelined ms per step)"
  [ -f gpurun_out/side_ab.log ] && grep "^\[" gpurun_out/side_ab.log | grep -v gpurun
  echo; echo "## dense worst case without spans: profiles/r06_dense_ab.txt"
} > profiles/r06_ubench.txt
python - <<'PY'
import json
d = json.loads(open('profiles/r06_bench.json').read().strip().splitlines()[-1])
print('stale', d['roofline'].get('counters_stale'), 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])
PY
