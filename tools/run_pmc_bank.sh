cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bank
mkdir -p $OUT
export DDSPP_NO_SIDE_STREAM=1
CMD="python $GRAFT_REPO_ROOT/tools/trace_case.py ${1:-headline} dict 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c -- $CMD > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_pmc_summary.py $OUT 2>&1 | head -120
