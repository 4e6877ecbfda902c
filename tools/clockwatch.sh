#!/bin/bash
# Development aid: sample clocks and power while the bench runs (is the kernel clock- or power-limited?)
# usage: clockwatch.sh "<bench args>"   (env is inherited: SDR_HIP_LIB / SDR_DIAG_SKIP select diagnostic builds)
python bench.py --no-cpu-baseline --steps ${STEPS:-60000} --warmup 10 $1 > gpurun_out/clockwatch_bench.log 2>&1 &
BP=$!
sleep 20
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '
  echo
  sleep 0.5
done
wait $BP
grep -o '"ms_per_step": [0-9.]*' gpurun_out/clockwatch_bench.log
