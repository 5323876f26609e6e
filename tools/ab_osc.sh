#!/bin/bash
# same-box A/B of the graded kernel (ddspp_cos_oscillator_bank, materialised envelopes, spans = 1):
# tools/ab_osc.sh <tag> <lib tag> [<lib tag> ...]   (libraries built by tools/build_variant.py)  -> gpurun_out/<tag>/ab.txt
cd $GRAFT_REPO_ROOT
TAG=${1:-abo}; shift
mkdir -p gpurun_out/$TAG
{
for i in 1 2; do
  echo -n "HEAD   "; python tools/bench_kernels.py --which osc --spans 1 --reps 4 2>&1 | grep -i "osc" | head -2 | tr '\n' ' '; echo
  for v in "$@"; do
    echo -n "$v  "; DDSPP_LIB=$GRAFT_REPO_ROOT/ddsp_piano_amd/libddspp_$v.so python tools/bench_kernels.py --which osc --spans 1 --reps 4 2>&1 | grep -i "osc" | head -2 | tr '\n' ' '; echo
  done
done
} > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
