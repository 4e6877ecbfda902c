#!/bin/bash
# pipelined step time under different stream plans (diagnostic library).  Plan = stream (0-5) of each kernel:
# fft, window means, noise stats, thresholds, gather, cumulate, find peaks, decode.   usage: ab_plan.sh [ENV=..] plan ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SDR_HIP_LIB=$PWD/tools/abl/libdiag.so
for plan in "$@"; do
  p=$(SDR_DIAG_PLAN=$plan timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1000 --warmup 100 2>&1 | grep -o '"ms_per_step": [0-9.]*' | awk '{print $2}')
  echo "GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-default} plan $plan: pipelined_step=$p ms"
done
