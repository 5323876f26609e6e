#!/bin/bash
# sclk / socket power while a kernel loops: is the chip power-limited under the VALU-bound kernels?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/power
probe() {  # name, command...
  name=$1; shift
  "$@" > /dev/null 2>&1 &
  pid=$!
  sleep ${WARM:-6}
  for i in 1 2 3; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
    sleep 1
  done | sed "s/^/$name: /"
  kill $pid 2>/dev/null; wait $pid 2>/dev/null
}
{
WARM=2 probe idle sleep 8
WARM=3 probe walk3 ./tools/ubench/walk_step 0 400000
WARM=3 probe walk1 ./tools/ubench/walk_step 110000 400000
WARM=14 probe noise python tools/noise_case.py 8 40000
WARM=14 probe bank python tools/bank_time.py headline 30000
WARM=14 probe step python tools/trace_case.py headline dict 12000
WARM=16 probe graded python tools/bench_kernels.py --which osc --reps 600
} 2>&1 | tee gpurun_out/power/probe.txt
