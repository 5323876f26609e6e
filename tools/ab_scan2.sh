#!/bin/bash
# round 5: wavefronts per SIMD of bank_scan_kernel (in-tree 4; libddspp_wpe5.so / _wpe8.so: 5 / 8), and the compacted scan at the
# two-sub-string and 192-harmonic shapes (vibrato on every voice)  -> gpurun_out/<tag>/ab.txt
cd $GRAFT_REPO_ROOT
TAG=${1:-abscan6}
mkdir -p gpurun_out/$TAG
L=$GRAFT_REPO_ROOT/ddsp_piano_amd
{
for i in 1 2; do
  for c in moving dense; do
    echo "OLD      bank $c $(DDSPP_OSC_COMPACT_SCAN=0 python tools/bank_time.py $c 20 2>/dev/null | tail -1)"
    echo "SCAN wpe4 bank $c $(python tools/bank_time.py $c 20 2>/dev/null | tail -1)"
    echo "SCAN wpe5 bank $c $(DDSPP_LIB=$L/libddspp_wpe5.so python tools/bank_time.py $c 20 2>/dev/null | tail -1)"
    echo "SCAN wpe8 bank $c $(DDSPP_LIB=$L/libddspp_wpe8.so python tools/bank_time.py $c 20 2>/dev/null | tail -1)"
  done
  for c in dafx24moving enst32kmoving; do
    echo "OLD  $c $(DDSPP_OSC_COMPACT_SCAN=0 python tools/trace_case.py $c dict 10 2>/dev/null | tail -1)"
    echo "SCAN $c $(python tools/trace_case.py $c dict 10 2>/dev/null | tail -1)"
  done
done
} > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_fuzz.py tests/test_gpu_shipped_configs.py -x -q -m gpu 2>&1 | tail -3
