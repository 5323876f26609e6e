cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_serial
mkdir -p $OUT
export DDSPP_NO_SIDE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c -- python $GRAFT_REPO_ROOT/tools/trace_case.py ${1:-headline} ${2:-dict} 10 > $OUT/kt.log 2>&1
tail -1 $OUT/kt.log
python - <<PY
import csv,glob
f=glob.glob('$OUT/kt/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=0
for r in rows[:30]:
    if 'at::native' in r['Name'] and int(r['Calls'])<10: continue
    per_step=float(r['TotalDurationNs'])/12/1e3
    tot+=per_step
    print('%-86s %5s %8.1f us avg %8.1f us/step' % (r['Name'][:86], r['Calls'], float(r['AverageNs'])/1e3, per_step))
print('sum', tot)
PY
