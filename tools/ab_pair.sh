#!/bin/bash
# same-box A/B of the paired sub-strings in the compacted bank (round 5): DDSPP_OSC_PAIR=0 against the default, base library
# against the in-tree build, at the two-sub-string shapes; then the tests that cover the bank.  -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
TAG=${1:-abpair}
mkdir -p gpurun_out/$TAG
BASE=$GRAFT_REPO_ROOT/ddsp_piano_amd/libddspp_base.so
{
for i in 1 2 3; do
  for c in dafx24 dafx22; do
    echo -n "BASE    $c "; DDSPP_LIB=$BASE python tools/trace_case.py $c dict 20 | tail -1
    echo -n "NOPAIR  $c "; DDSPP_OSC_PAIR=0 python tools/trace_case.py $c dict 20 | tail -1
    echo -n "PAIR    $c "; python tools/trace_case.py $c dict 20 | tail -1
  done
done
for i in 1 2; do
  echo -n "BASE step "; DDSPP_LIB=$BASE python tools/trace_case.py headline dict 20 | tail -1
  echo -n "NEW  step "; python tools/trace_case.py headline dict 20 | tail -1
done
} > gpurun_out/$TAG/ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/$TAG/ab.txt
timeout 1500 python -m pytest tests/test_gpu_osc.py tests/test_gpu_group.py tests/test_gpu_shipped_configs.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
tail -8 gpurun_out/$TAG/pytest.log
bash tools/gpu_sysfs_probe.sh > gpurun_out/$TAG/sysfs.txt 2>&1
