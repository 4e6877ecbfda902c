# usage: ab_env.sh "ENV1=a ENV2=b" "ENV..." ...   (each argument = one configuration's environment)
for i in 1 2; do
for cfg in "$@"; do
  s=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --kernel-breakdown --serial 2>&1 | grep -E "^ +k_fft" | awk '{print $2}')
  p=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*' | awk '{print $2}')
  echo "[$cfg] fft_serial=$s ms  pipelined_step=$p ms"
done; done
