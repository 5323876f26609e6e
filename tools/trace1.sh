# per-kernel breakdown of one bench input case without the side stream (clean kernel durations): trace1.sh <case> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
c=${1:-moving}; TAG=${2:-t1}; FORM=${3:-dict}
mkdir -p $R/gpurun_out/$TAG
DDSPP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/$c -o $c -- python $R/tools/trace_case.py $c $FORM 10 > $R/gpurun_out/$TAG/$c.log 2>&1
grep "ms per step" $R/gpurun_out/$TAG/$c.log
f=$(find $R/gpurun_out/$TAG/$c -name "*kernel_stats.csv" | head -1)
python - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print('%-90s calls %5s avg_us %9.1f pct %5s' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
find $R/gpurun_out/$TAG/$c -name "*.csv" ! -name "*kernel_stats.csv" -delete
