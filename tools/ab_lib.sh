#!/bin/bash
# usage: [R=3] [SERIAL=1 KGREP=regex] [ENVS="A=1 B=2"] ab_lib.sh name1 name2 ...
# ("cur" = the in-tree library, otherwise tools/abl/lib<name>.so, built by build_abl.sh).  Pipelined step of each
# library, R interleaved rounds on the same box (boxes differ by several per cent, and so do runs on a busy host:
# compare minima of the same call only); with SERIAL=1 the standalone kernel times too.
cd ${GRAFT_REPO_ROOT:-/root/repo}
declare -A all
for i in $(seq 1 ${R:-3}); do
for l in "$@"; do
  if [ "$l" = cur ]; then unset SDR_HIP_LIB; else export SDR_HIP_LIB=$PWD/tools/abl/lib$l.so; fi
  p=$(env $ENVS timeout -k 10 200 python bench.py --no-cpu-baseline --steps ${STEPS:-1500} --warmup 150 2>&1 | grep -o '"ms_per_step": [0-9.]*' | awk '{print $2}')
  all[$l]="${all[$l]} $p"
  if [ -n "$SERIAL" ] && [ $i = 1 ]; then echo "$l:"; timeout -k 10 200 python bench.py --no-cpu-baseline --kernel-breakdown --serial --steps 300 --warmup 30 2>&1 | grep -E "^ +k_${KGREP:-(cum|fft)}"; fi
done; done
for l in "$@"; do
  echo "$l: pipelined_step ms:${all[$l]}  min $(echo ${all[$l]} | tr ' ' '\n' | sort -n | head -1)"
done
