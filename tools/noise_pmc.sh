#!/bin/bash
# SQ / LDS counters of the FilteredNoise kernel alone (tools/noise_case.py): noise_pmc.sh <tag> [vq]; extra environment
# (DDSPP_FIR_WIN=0, DDSPP_WIN_DEBUG=..) is inherited.  -> gpurun_out/<tag>/summary.txt
set -u
TAG=${1:-npmc}; VQ=${2:-8}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/noise_case.py $VQ 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c -- $CMD > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_VALU_FMA_F32 --output-format csv -d $OUT/c -o p -- $CMD > $OUT/c.log 2>&1
cd $R
python tools/step_pmc_summary.py $OUT 2>&1 | grep -A45 "noise_\|tv_fir" > $OUT/summary.txt
cat $OUT/summary.txt
find $OUT -name "*.csv" ! -name "*kernel_stats.csv" -delete
