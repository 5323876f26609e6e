cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_cases
mkdir -p $OUT
for c in headline moving dense; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c -o t -- python $GRAFT_REPO_ROOT/tools/trace_case.py $c dict 5 > $OUT/$c.log 2>&1
  tail -1 $OUT/$c.log
  python - <<PY
import csv,glob
f=glob.glob('$OUT/$c/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:9]:
    print('%-90s %5s %10.1f us avg' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
