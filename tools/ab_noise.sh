#!/bin/bash
# same-box A/B of the FilteredNoise kernel: ddsp_piano_amd/libddspp_base.so (DDSPP_LIB) against the in-tree build;
# then the noise / group tests on the new build.  -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
TAG=${1:-abn}
mkdir -p gpurun_out/$TAG
BASE=$GRAFT_REPO_ROOT/ddsp_piano_amd/libddspp_base.so
{
for i in 1 2; do
  echo "== BASE"; DDSPP_LIB=$BASE python tools/bench_kernels.py --which noisev --reps 10 2>&1 | grep "vq=8"
  echo "== NEW";  python tools/bench_kernels.py --which noisev --reps 10 2>&1 | grep "vq=8"
done
for c in headline c5 enst32k dafx22 enst8k; do
  echo -n "BASE "; DDSPP_LIB=$BASE python tools/trace_case.py $c dict 20 | tail -1
  echo -n "NEW  "; python tools/trace_case.py $c dict 20 | tail -1
done
} > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
timeout 1500 python -m pytest tests/test_gpu_noise_reverb.py tests/test_gpu_group.py tests/test_gpu_full_size.py tests/test_gpu_enstdkcl.py -x -q -m gpu > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
tail -15 gpurun_out/$TAG/pytest.log
