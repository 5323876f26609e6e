cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02d/pytest.log
tail -15 gpurun_out/r02d/pytest.log
python tools/bench_kernels.py --which compact,controls,noise --reps 10 2>&1 | tail -6
DDSPP_NO_SIDE_STREAM=1 python tools/trace_case.py headline dict 10 2>&1 | tail -1
