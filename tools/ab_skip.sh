# pipelined step time with pipeline stages left out (diagnostic library; results are wrong by construction)
export SDR_HIP_LIB=$PWD/tools/abl/libdiag.so
for m in 0 2 128 130 254; do
  p=$(SDR_DIAG_SKIP=$m timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*' | awk '{print $2}')
  echo "skip mask $m: pipelined_step=$p ms"
done
