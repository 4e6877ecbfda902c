for r in 1 2 3; do
for e in "SDR_PARK_FIRST=0" "SDR_PARK_FIRST=1"; do
  echo -n "$e long: "; env $e timeout -k 10 200 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  echo -n "$e s20: "; env $e timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done
