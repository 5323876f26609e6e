cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/bank_trace
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c -- python $GRAFT_REPO_ROOT/tools/bank_time.py headline 5 > $OUT/kt.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('$OUT/kt/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:8]:
    print('%-100s %5s %10.1f us avg' % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3))
PY
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- python $GRAFT_REPO_ROOT/tools/bank_time.py headline 3 > $OUT/a.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('$OUT/a/**/*counter_collection.csv', recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,c in agg.items():
    if 'bank_compact' in k or 'prepass' in k:
        print(k); [print('   %-20s %16.0f' % (n, sum(v)/len(v))) for n,v in sorted(c.items())]
PY
