#!/bin/bash
# The moving_f0 / dense_worst_case inputs of bench.py, kernel by kernel: rocprofv3 --kernel-trace --stats over
# tools/trace_case.py with the memoised and the chunk-parallel pre-pass (DDSPP_OSC_PLAIN_PREPASS=1), then the SQ counters
# of the moving case (separate --pmc passes).  usage: tools/moving_pmc.sh [tag]  -> gpurun_out/<tag>/summary.txt
set -u
TAG=${1:-moving_pmc}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for CASE in headline moving dense; do
  for V in memo plain; do
    if [ $V = plain ]; then export DDSPP_OSC_PLAIN_PREPASS=1; else unset DDSPP_OSC_PLAIN_PREPASS; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${CASE}_$V -o c -- python $R/tools/trace_case.py $CASE dict 8 > $O/kt_${CASE}_$V.log 2>&1
    python - <<PY >> $O/summary.txt
import csv,glob
f=glob.glob('$O/kt_${CASE}_$V/**/*kernel_stats.csv',recursive=True)
print('== $CASE $V', [l for l in open('$O/kt_${CASE}_$V.log') if 'ms per step' in l][-1].strip())
for r in list(csv.DictReader(open(f[0])))[:10]:
    print('   %-80s %5s %10.1f us %5s%%' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
  done
done
unset DDSPP_OSC_PLAIN_PREPASS
CMD="python $R/tools/trace_case.py moving dict 3"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $O/a -o p -- $CMD > $O/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b -o p -- $CMD > $O/b.log 2>&1
cd $R
cp -r $O/kt_moving_memo $O/kt
python tools/step_pmc_summary.py $O 2>&1 | grep -A30 "osc_prepass\|bank_compact\|offset_scan" >> $O/summary.txt
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete
cat $O/summary.txt
