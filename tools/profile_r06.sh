#!/bin/bash
# Round-6 profile on the GPU box, ONE pass on the final tree: kernel-trace stats of bench.py (default command) with the JSON line
# printed inside that run, HBM counters of the graded kernel (-> osc_traffic.json), SQ / HBM counters of the step's kernels
# (tools/step_pmc.sh -> step_valu.json), the graded kernel in eight processes (placement spread), the round's probes.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r06
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python $R/bench.py --steps 10 --warmup 2 --sustain-seconds 1 > $OUT/bench_under_rocprof.log 2>&1
grep "^{\"metric" $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python $R/tools/bench_kernels.py --which osc --spans 1 --reps 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python $R/tools/bench_kernels.py --which osc --spans 1 --reps 2 > $OUT/pmc_write.log 2>&1
cd $R
python tools/osc_traffic.py $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1) $OUT/osc_traffic.json
{
  python tools/pmc_summary.py $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) "osc_stream_kernel<2"
  python tools/pmc_summary.py $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1) "osc_stream_kernel<2"
  python tools/pmc_summary.py $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) "resample_"
  python tools/pmc_summary.py $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1) "resample_"
} > $OUT/osc_pmc.txt 2>&1
bash tools/step_pmc.sh r06 > $OUT/step_pmc.log 2>&1
cp gpurun_out/step_pmc_r06/summary.txt $OUT/step_pmc_summary.txt
cp gpurun_out/step_pmc_r06/step_valu.json $OUT/step_valu.json
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv
# the bench line with the fresh counters in place
cp $OUT/step_valu.json $OUT/osc_traffic.json profiles/
python bench.py > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*.csv" ! -name "bench_kernel_stats.csv" -delete
# the graded kernel, old and new, in eight processes one after the other: the placement lottery (DESIGN_LOG 4a)
for i in 1 2 3 4 5 6 7 8; do
  ./tools/ubench/osc_graded 1024 3 2>&1 | grep -E "rounds 1-5|library route|bare read|ABL=1|ABL=4" | head -5 | sed "s/^/process $i: /"
done > $OUT/osc_graded_spread.txt
./tools/ubench/osc_graded 1024 5 > $OUT/osc_graded.txt 2>&1
./tools/ubench/lds_stream > $OUT/lds_stream.txt 2>&1
./tools/ubench/store_cost > $OUT/store_cost.txt 2>&1
./tools/ubench/philox_rates > $OUT/philox_rates.txt 2>&1
python tools/resample_time.py > $OUT/resample_time.txt 2>&1
ls $OUT
