cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fir_pmc
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --which fir --reps 2 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --which fir --reps 2 > $OUT/b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/a/p_counter_collection.csv fir_
python tools/pmc_summary.py $OUT/b/p_counter_collection.csv fir_
