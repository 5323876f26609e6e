#!/bin/bash
# Round-5 profile on the GPU box: kernel-trace stats of bench.py (default command) with the JSON line printed inside that
# run, SQ / HBM counters of the step's kernels (tools/step_pmc.sh -> step_valu.json), HBM counters of the graded kernel
# (-> osc_traffic.json, both carry the hash of the kernel sources), the FilteredNoise kernel alone (tools/noise_pmc.sh).
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r05
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python $R/bench.py --steps 10 --warmup 2 --sustain-seconds 1 > $OUT/bench_under_rocprof.log 2>&1
grep "^{\"metric" $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python $R/tools/bench_kernels.py --which osc --spans 1 --reps 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python $R/tools/bench_kernels.py --which osc --spans 1 --reps 2 > $OUT/pmc_write.log 2>&1
cd $R
python tools/osc_traffic.py $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1) $OUT/osc_traffic.json
python tools/pmc_summary.py $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) "osc_kernel<1, false" > $OUT/osc_pmc.txt
python tools/pmc_summary.py $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1) "osc_kernel<1, false" >> $OUT/osc_pmc.txt
bash tools/step_pmc.sh r05 > $OUT/step_pmc.log 2>&1
cp gpurun_out/step_pmc_r05/summary.txt $OUT/step_pmc_summary.txt
cp gpurun_out/step_pmc_r05/step_valu.json $OUT/step_valu.json
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv
bash tools/noise_pmc.sh prof_r05_noise 8 > /dev/null 2>&1
cp gpurun_out/prof_r05_noise/summary.txt $OUT/noise_pmc.txt
# the bench line with the fresh counters in place
cp $OUT/step_valu.json $OUT/osc_traffic.json profiles/
python bench.py > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*.csv" ! -name "bench_kernel_stats.csv" -delete
ls $OUT
# round 5: what the chip sustains in float32 multiply-adds, the walk of the FilteredNoise kernel alone, socket power
./tools/ubench/fma_ceiling 20 > $OUT/fma_ceiling.txt 2>&1
for pad in 0 20000 110000; do ./tools/ubench/walk_step $pad 400 | tail -1; done > $OUT/walk_step.txt 2>&1
bash tools/power_probe.sh > /dev/null 2>&1
cp gpurun_out/power/probe.txt $OUT/power_probe.txt
ls $OUT
