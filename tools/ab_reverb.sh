#!/bin/bash
# reverb kernels under the kernel trace (base library through DDSPP_LIB against the in-tree build), then the reverb tests
cd $GRAFT_REPO_ROOT
TAG=${1:-abr}
for v in base new; do
  if [ $v = base ]; then export DDSPP_LIB=$GRAFT_REPO_ROOT/ddsp_piano_amd/libddspp_base.so; else unset DDSPP_LIB; fi
  for c in headline c5; do
    echo "== $v $c"; bash tools/trace_step.sh ${TAG}_${v}_$c $c 2>&1 | grep -E "part_|ms per step"
  done
done
unset DDSPP_LIB
timeout 1500 python -m pytest tests/test_gpu_noise_reverb.py tests/test_gpu_reverb_models.py tests/test_gpu_group.py tests/test_gpu_full_size.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -3
