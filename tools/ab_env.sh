#!/bin/bash
# usage: [R=3] ab_env.sh "ENV1=a ENV2=b" "ENV..." ...   (each argument = one configuration's environment; "-" = none)
# pipelined step time, R interleaved rounds on the same box, minimum last
cd ${GRAFT_REPO_ROOT:-/root/repo}
declare -A all
for i in $(seq 1 ${R:-3}); do
for cfg in "$@"; do
  e="$cfg"; [ "$cfg" = "-" ] && e=""
  p=$(env $e timeout -k 10 200 python bench.py --no-cpu-baseline --steps ${STEPS:-1500} --warmup 150 2>&1 | grep -o '"ms_per_step": [0-9.]*' | awk '{print $2}')
  all[$cfg]="${all[$cfg]} $p"
done; done
for cfg in "$@"; do
  echo "[$cfg] pipelined_step ms:${all[$cfg]}  min $(echo ${all[$cfg]} | tr ' ' '\n' | sort -n | head -1)"
done
