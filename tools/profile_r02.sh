#!/bin/bash
# Round-2 profile on the GPU box: kernel-trace stats of bench.py (default command), the JSON line printed inside that
# run, SQ / HBM counters of the step's kernels (tools/step_pmc.sh), HBM counters of the graded kernel.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 > $OUT/bench_under_rocprof.log 2>&1
grep "^{\"metric" $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --which osc --spans 1 --reps 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --which osc --spans 1 --reps 2 > $OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) "osc_kernel<1, false" > $OUT/osc_pmc.txt
python tools/pmc_summary.py $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1) "osc_kernel<1, false" >> $OUT/osc_pmc.txt
bash tools/step_pmc.sh r02 > $OUT/step_pmc.log 2>&1
cp gpurun_out/step_pmc_r02/summary.txt $OUT/step_pmc_summary.txt
cp gpurun_out/step_pmc_r02/step_valu.json $OUT/step_valu.json
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv
python bench.py > $OUT/bench.json 2> $OUT/bench.err
ls $OUT
