#!/bin/bash
# Development aid: CU-time of every kernel (SQ_BUSY_CU_CYCLES = cycles with a wave resident, summed over CUs):
# what each tail stage takes away from the FFT, whose workgroups need whole CUs.
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 120 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d gpurun_out/pmc_cutime -o c --output-format csv -- python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --serial > gpurun_out/pmc_cutime.log 2>&1 || { echo failed; exit 1; }
python3 - <<'PY'
import csv, collections, glob
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_cutime/**/c_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        # (k_noise_exact_check belongs to the record read behind the run, not to a step)
        if "sdr::" in r["Kernel_Name"] and "k_noise_exact_check" not in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sdr::", "")
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
ks = sorted({k for k, _ in agg})
tot = 0
for k in ks:
    cu = sum(agg[(k, "SQ_BUSY_CU_CYCLES")]) / len(agg[(k, "SQ_BUSY_CU_CYCLES")])
    w = sum(agg[(k, "SQ_WAVES")]) / len(agg[(k, "SQ_WAVES")])
    tot += cu
    print(f"{k:24s} {cu/2.4e9*1e3:8.2f} CU-ms   {w:9.0f} waves")
print(f"{'sum':24s} {tot/2.4e9*1e3:8.2f} CU-ms  (a 0.688 ms step is 176 CU-ms)")
PY
