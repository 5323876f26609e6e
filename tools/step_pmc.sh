#!/bin/bash
# SQ counters of the kernels of one timed step (separate --pmc passes over a short bench run, branches serialised so
# that per-kernel numbers mean something).  usage: tools/step_pmc.sh <tag>  -> gpurun_out/step_pmc_<tag>/summary.txt
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/step_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DDSPP_NO_SIDE_STREAM=1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-extras --sustain-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c -- $CMD > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/c -o p -- $CMD > $OUT/c.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o p -- $CMD > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o p -- $CMD > $OUT/w.log 2>&1
cd $GRAFT_REPO_ROOT
{
  echo "# tools/step_pmc.sh $TAG: $CMD (DDSPP_NO_SIDE_STREAM=1), separate --pmc passes; per-launch averages"
  python tools/step_pmc_summary.py $OUT
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
