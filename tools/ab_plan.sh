#!/bin/bash
# pipelined step time under different stream plans (diagnostic library).  Plan = stream of each kernel:
# fft, window means, noise stats, thresholds, gather, cumulate, find peaks, decode
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SDR_HIP_LIB=$PWD/tools/abl/libdiag.so
for plan in "$@"; do
  p=$(SDR_DIAG_PLAN=$plan timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1000 --warmup 100 2>&1 | grep -o '"ms_per_step": [0-9.]*' | awk '{print $2}')
  echo "plan $plan: pipelined_step=$p ms"
done
