#!/bin/bash
# Development aid: memory-side counters of k_fft_psd (one rocprofv3 pass per group, no trace domains).
# (A group of TA_* counters made rocprofv3 abort and the run hang on this pool: left out.)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d gpurun_out/pmc_mem -o m$i --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial > gpurun_out/pmc_mem_m$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/pmc_mem/**/m*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_fft_psd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:40s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
