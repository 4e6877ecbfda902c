#!/bin/bash
# pipelined step time under different stream plans (diagnostic library).  Plan = stream (0-3) of each kernel:
# fft, window means, noise stats, thresholds, gather, cumulate, find peaks, decode.
# usage: [R=3] [LIB=diag] [BENCH_ARGS="--workload c5"] [STEPS=600] ab_plan.sh plan ...     (interleaved rounds on one box, minimum last)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SDR_HIP_LIB=$PWD/tools/abl/lib${LIB:-diag}.so
declare -A all
for i in $(seq 1 ${R:-3}); do
for plan in "$@"; do
  p=$(SDR_DIAG_PLAN=$plan timeout -k 10 200 python bench.py $BENCH_ARGS --no-cpu-baseline --steps ${STEPS:-1500} --warmup 150 2>&1 | grep -o '"ms_per_step": [0-9.]*' | awk '{print $2}')
  all[$plan]="${all[$plan]} $p"
done; done
for plan in "$@"; do
  echo "${LIB:-diag} plan $plan: pipelined_step ms:${all[$plan]}  min $(echo ${all[$plan]} | tr ' ' '\n' | sort -n | head -1)"
done
