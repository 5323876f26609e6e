#!/bin/bash
# same-box A/B of get_controls: the all-purpose kernel (DDSPP_CONTROLS_GENERIC=1) against the lean one, then the tests
# that cover it.  -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
TAG=${1:-abc}
mkdir -p gpurun_out/$TAG
{
for i in 1 2; do
  echo "== GENERIC"; DDSPP_CONTROLS_GENERIC=1 python tools/bench_kernels.py --which controls --reps 10 2>&1 | grep "inharmonic_controls"
  echo "== LEAN";    python tools/bench_kernels.py --which controls --reps 10 2>&1 | grep "inharmonic_controls"
done
for c in headline c5 dafx22; do
  echo -n "GENERIC "; DDSPP_CONTROLS_GENERIC=1 python tools/trace_case.py $c dict 20 2>/dev/null | tail -1
  echo -n "LEAN    "; python tools/trace_case.py $c dict 20 2>/dev/null | tail -1
done
} > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
timeout 1500 python -m pytest tests/test_gpu_osc.py tests/test_gpu_group.py tests/test_gpu_golden.py tests/test_gpu_enstdkcl.py -x -q -m gpu > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
tail -15 gpurun_out/$TAG/pytest.log
