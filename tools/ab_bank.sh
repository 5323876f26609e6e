#!/bin/bash
# same-box A/B of the compacted bank: ddsp_piano_amd/libddspp_base.so (DDSPP_LIB) against the in-tree build, then the
# oscillator / group tests on the new build.  -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
TAG=${1:-abb}
mkdir -p gpurun_out/$TAG
BASE=$GRAFT_REPO_ROOT/ddsp_piano_amd/libddspp_base.so
{
for i in 1 2 3; do
  for c in headline; do
    echo -n "BASE $c "; SPLIT_LAST=1 DDSPP_LIB=$BASE python tools/bank_time.py $c 30 2>&1 | tail -1
    echo -n "NEW  $c "; SPLIT_LAST=1 python tools/bank_time.py $c 30 2>&1 | tail -1
  done
done
for c in moving dense; do
  echo -n "BASE $c "; SPLIT_LAST=1 DDSPP_LIB=$BASE python tools/bank_time.py $c 10 2>&1 | tail -1
  echo -n "NEW  $c "; SPLIT_LAST=1 python tools/bank_time.py $c 10 2>&1 | tail -1
done
for i in 1 2; do
  echo -n "BASE step "; DDSPP_LIB=$BASE python tools/trace_case.py headline dict 20 | tail -1
  echo -n "NEW  step "; python tools/trace_case.py headline dict 20 | tail -1
done
echo -n "BASE c5 "; DDSPP_LIB=$BASE python tools/trace_case.py c5 dict 10 | tail -1
echo -n "NEW  c5 "; python tools/trace_case.py c5 dict 10 | tail -1
} > gpurun_out/$TAG/ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/$TAG/ab.txt
timeout 1500 python -m pytest tests/test_gpu_osc.py tests/test_gpu_group.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
tail -5 gpurun_out/$TAG/pytest.log
