# counters of the compacted additive path (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/compact_pmc
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/bench_kernels.py --which compact --reps 2"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/a/p_counter_collection.csv osc_
python tools/pmc_summary.py $OUT/b/p_counter_collection.csv osc_
tail -3 $OUT/a.log
