#!/bin/bash
# pipelined step time with pipeline stages left out (diagnostic library; results are wrong by construction)
# mask bits: 1 fft, 2 window means, 4 noise stats, 8 thresholds, 16 gather, 32 cumulate, 64 find peaks, 128 decode
# usage: [LIB=diag] ab_skip.sh mask ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SDR_HIP_LIB=$PWD/tools/abl/lib${LIB:-diag}.so
for m in "$@"; do
  p=$(SDR_DIAG_SKIP=$m timeout -k 10 200 python bench.py --no-cpu-baseline --no-delivery --steps 1000 --warmup 100 2>&1 | grep -o '"ms_per_step": [0-9.]*' | awk '{print $2}')
  echo "${LIB:-diag} skip mask $m: pipelined_step=$p ms"
done
