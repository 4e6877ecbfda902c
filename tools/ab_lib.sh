#!/bin/bash
# usage: ab_lib.sh name1 name2 ...   ("cur" = the in-tree library, otherwise tools/abl/lib<name>.so); pipelined step
# and, with SERIAL=1, the standalone kernel times too
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
for l in "$@"; do
  if [ "$l" = cur ]; then unset SDR_HIP_LIB; else export SDR_HIP_LIB=$PWD/tools/abl/lib$l.so; fi
  p=$(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1000 --warmup 100 2>&1 | grep -o '"ms_per_step": [0-9.]*' | awk '{print $2}')
  echo "$l: pipelined_step=$p ms"
  if [ -n "$SERIAL" ] && [ $i = 1 ]; then timeout -k 10 200 python bench.py --no-cpu-baseline --kernel-breakdown --serial --steps 300 --warmup 30 2>&1 | grep -E "^ +k_(cum|fft)"; fi
done; done
