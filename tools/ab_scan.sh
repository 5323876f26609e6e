#!/bin/bash
# same-box A/B of the compacted scan of moving chunks (round 5): DDSPP_OSC_COMPACT_SCAN=0 against the default on the
# moving / dense / headline inputs, then the tests that cover the span starts.  -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
TAG=${1:-abscan}
mkdir -p gpurun_out/$TAG
{
timeout 900 python -m pytest tests/test_gpu_group.py -x -q -m gpu -k "compacted_scan or moving_frequencies or paired" 2>&1 | tail -3
for i in 1 2; do
  for c in moving dense headline; do
    echo "OLD   bank $c $(DDSPP_OSC_COMPACT_SCAN=0 python tools/bank_time.py $c 20 2>/dev/null | tail -1)"
    echo "SCAN2 bank $c $(python tools/bank_time.py $c 20 2>/dev/null | tail -1)"
    echo "SCAN1 bank $c $(DDSPP_OSC_SCAN_VPL=1 python tools/bank_time.py $c 20 2>/dev/null | tail -1)"
    echo "SCAN2/8k bank $c $(DDSPP_OSC_SCAN_WAVES=8192 python tools/bank_time.py $c 20 2>/dev/null | tail -1)"
  done
done
for c in moving headline; do
  echo "OLD  $c $(DDSPP_OSC_COMPACT_SCAN=0 python tools/trace_case.py $c dict 20 2>/dev/null | tail -1)"
  echo "SCAN $c $(python tools/trace_case.py $c dict 20 2>/dev/null | tail -1)"
done
} > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
if [ "${2:-}" = "tests" ]; then
timeout 1500 python -m pytest tests/test_gpu_osc.py tests/test_gpu_group.py tests/test_gpu_fuzz.py tests/test_gpu_streaming.py tests/test_gpu_long_file.py tests/test_gpu_shipped_configs.py -x -q -m gpu > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
tail -8 gpurun_out/$TAG/pytest.log
fi
