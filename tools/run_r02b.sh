cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02b/pytest.log
tail -30 gpurun_out/r02b/pytest.log
for c in headline moving dense; do timeout 300 python tools/trace_case.py $c dict 10 2>&1 | tail -1; done
timeout 300 python tools/trace_case.py headline audio 10 2>&1 | tail -1
