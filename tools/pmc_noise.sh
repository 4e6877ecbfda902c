#!/bin/bash
# Development aid: LDS-side counters of the noise-floor kernels (one rocprofv3 pass per group)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
i=0
for grp in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp -d gpurun_out/pmc_noise -o n$i --output-format csv -- python3 bench.py --steps 2 --warmup 1 --settle-ms 0 --no-cpu-baseline --serial > gpurun_out/pmc_noise_n$i.log 2>&1 || { echo "group $i failed"; exit 1; }
done
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/pmc_noise/**/n*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for k in ("k_noise_stats", "k_window_means"):
            if k in r["Kernel_Name"]:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print(f"{k:16s} {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
