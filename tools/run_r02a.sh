cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
tail -3 gpurun_out/r02a/pytest.log
timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r02a/bench.json
timeout 900 bash tools/step_pmc.sh r02a > gpurun_out/r02a/pmc.log 2>&1
tail -50 gpurun_out/r02a/pmc.log
