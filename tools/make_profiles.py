#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs a gpurun call left under gpurun_out/ into the tracked summaries under
profiles/ (kernel stats of the sdr:: kernels, PMC traffic per launch, the bench JSON line).

Commands that produced the inputs (run on the GPU box, from the repo root):
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01 -o r01 --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial
  rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial
  python bench.py --steps 40 --warmup 5 > gpurun_out/bench_full.log
  python bench.py --steps 10 --warmup 3 --kernel-breakdown --no-cpu-baseline --serial 2> gpurun_out/bench_serial.err
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
key = sys.argv[2] if len(sys.argv) > 2 else "c3_f2048"

rows = list(csv.reader(open(os.path.join(G, "prof_r01", "r01_kernel_stats.csv"))))
out = [rows[0]] + [r for r in rows[1:] if "sdr::" in r[0]]
csv.writer(open(os.path.join(P, f"{tag}_kernel_stats_{key}.csv"), "w")).writerows(out)


def pmc(name):
    rows = list(csv.DictReader(open(os.path.join(G, f"pmc_{name}", f"{name}_counter_collection.csv"))))
    agg = collections.defaultdict(list)
    for r in rows:
        k = r["Kernel_Name"]
        if "sdr::" in k:
            agg[k.split("(")[0].replace("void ", "").replace("sdr::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


f, w = pmc("fetch"), pmc("write")
tab = {}
for k in f:
    tab[k] = {"FETCH_SIZE_KB_raw": round(f[k], 1), "WRITE_SIZE_KB_raw": round(w.get(k, 0), 1),
              "hbm_read_bytes_corrected": int(f[k] * 1024 * 2), "hbm_write_bytes": int(w.get(k, 0) * 1024),
              "hbm_bytes_per_launch": int(f[k] * 1024 * 2 + w.get(k, 0) * 1024)}
fft = [k for k in tab if k.startswith("k_fft_psd")][0]
tpath = os.path.join(P, "traffic.json")
doc = json.load(open(tpath)) if os.path.exists(tpath) else {}
doc["_how"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (no trace domains) over "
               "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial`; per-dispatch means. Units: KB. gfx950 "
               "correction per MI355X_MICROARCH.md §HBM: FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming read "
               "-> read bytes = FETCH_SIZE*1024*2 (checked here on k_cumulate, which streams exactly the 134.2 MB spectrum "
               "once); WRITE_SIZE is exact.")
doc[key] = {"k_fft_psd_hbm_bytes_per_launch": tab[fft]["hbm_bytes_per_launch"],
            "algorithmic_bytes_per_launch": 8 * 2048 * 16384, "kernels": tab}
json.dump(doc, open(tpath, "w"), indent=1)
shutil.copy(os.path.join(G, "bench_full.log"), os.path.join(P, f"{tag}_bench_{key}.json"))
shutil.copy(os.path.join(G, "bench_serial.err"), os.path.join(P, f"{tag}_kernel_breakdown_serial.txt"))
for r in out[1:]:
    print(r[0].split("(")[0].replace("void ", "")[:40], "calls", r[1], "avg_ns", r[3])
print(json.dumps(tab[fft]))
