#!/bin/bash
# usage: ab_env.sh "ENV1=a ENV2=b" "ENV..." ...   (each argument = one configuration's environment); pipelined step time
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  p=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1000 --warmup 100 2>&1 | grep -o '"ms_per_step": [0-9.]*' | awk '{print $2}')
  echo "[$cfg] pipelined_step=$p ms"
done
