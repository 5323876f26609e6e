# SQ counters of the kernels of one bench input case (no side stream): case_pmc.sh <case> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
c=${1:-moving}; TAG=${2:-cp}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export DDSPP_NO_SIDE_STREAM=1
CMD="python $R/tools/trace_case.py $c dict 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c -- $CMD > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/c -o p -- $CMD > $OUT/c.log 2>&1
cd $R
python tools/step_pmc_summary.py $OUT 2>&1 | grep -A30 "${3:-prepass_fused\|bank_compact_kernel}"
find $OUT -name "*.csv" ! -name "*kernel_stats.csv" -delete
