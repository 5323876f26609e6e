#!/bin/bash
# kernel list of one bench case under rocprofv3's kernel trace: tools/trace_step.sh <tag> [case] [ENV=val ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-trace}; CASE=${2:-headline}; shift; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o t -- python $R/tools/trace_case.py $CASE dict 20 > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/kt/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(out + '/kernels.txt', 'w') as w:
    for r in rows[:24]:
        line = f"{r['Name'][:86]:86s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {r['Percentage']}"
        print(line); w.write(line + '\n')
PY
rm -rf $OUT/kt
