# usage: ab_lib.sh name1 name2 ...   ("cur" = the in-tree library, otherwise tools/abl/lib<name>.so); three rounds
for i in 1 2 3; do
for l in "$@"; do
  if [ "$l" = cur ]; then unset SDR_HIP_LIB; else export SDR_HIP_LIB=$PWD/tools/abl/lib$l.so; fi
  s=$(timeout -k 10 200 python bench.py --no-cpu-baseline --kernel-breakdown --serial --steps 300 --warmup 30 2>&1 | grep -E "^ +k_fft" | awk '{print $2}')
  p=$(timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*' | awk '{print $2}')
  echo "$l: fft_standalone=$s ms  pipelined_step=$p ms"
done; done
