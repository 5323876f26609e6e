cd $GRAFT_REPO_ROOT
python tools/bench_kernels.py --which compact --reps 10 2>&1 | tail -1
for c in headline moving dense; do timeout 300 python tools/trace_case.py $c dict 10 2>&1 | tail -1; done
timeout 300 python tools/trace_case.py headline audio 10 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_group.py tests/test_gpu_osc.py -x -q 2>&1 | tail -3
