#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of bench.py, then HBM PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes) for the oscillator-bank kernel.  Outputs land in gpurun_out/prof_$1.
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --which osc --spans 1 --reps 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --which osc --spans 1 --reps 2 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --which osc,fused --spans 0 --reps 2 > $OUT/pmc_sq.log 2>&1
# the whole chain: HBM bytes per kernel (two more separate passes over a short bench run, same launches as a stats run)
# (DDSPP_NO_SIDE_STREAM=1: the two branches one after the other, so that per-kernel durations mean something)
export DDSPP_NO_SIDE_STREAM=1
CHAIN="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-single-stream"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain_kt -o c -- $CHAIN > $OUT/chain_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/chain_fetch -o c -- $CHAIN > $OUT/chain_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/chain_write -o c -- $CHAIN > $OUT/chain_write.log 2>&1
cd $GRAFT_REPO_ROOT
unset DDSPP_NO_SIDE_STREAM
python tools/chain_traffic.py $OUT/chain_kt/c_kernel_stats.csv $OUT/chain_fetch/c_counter_collection.csv $OUT/chain_write/c_counter_collection.csv > $OUT/chain_traffic.txt 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
ls -R $OUT | head -40
