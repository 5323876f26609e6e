cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -k config5 > gpurun_out/r02c/c5.log 2>&1
grep -n "Error\|error\|rc = \|FAILED\|passed\|failed" gpurun_out/r02c/c5.log | head -20
grep -n "polyphonic.py\|core.py" gpurun_out/r02c/c5.log | head
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/headline -o t -- python $GRAFT_REPO_ROOT/tools/trace_case.py headline dict 5 > $OUT/headline.log 2>&1
tail -1 $OUT/headline.log
python - <<PY
import csv,glob
f=glob.glob('$OUT/headline/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:12]:
    print('%-90s %5s %10.1f us avg' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
