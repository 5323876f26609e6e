#!/bin/bash
# matrix-pipe multiply-add ceiling (tools/ubench/mfma_ceiling) and the socket power / clock under it -> gpurun_out/mfma/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/mfma
./tools/ubench/mfma_ceiling 20 > gpurun_out/mfma/ceiling.txt 2>&1
probe() { name=$1; shift; "$@" > /dev/null 2>&1 & pid=$!; sleep 4; for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 1; done | sed "s/^/$name: /"; kill $pid 2>/dev/null; wait $pid 2>/dev/null; }
probe mfma_lds ./tools/ubench/mfma_ceiling 20 1 >> gpurun_out/mfma/ceiling.txt 2>&1
probe vfma ./tools/ubench/mfma_ceiling 20 2 >> gpurun_out/mfma/ceiling.txt 2>&1
cat gpurun_out/mfma/ceiling.txt
