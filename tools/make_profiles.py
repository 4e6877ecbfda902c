#!/usr/bin/env python3
"""Condenses what tools/run_profiles.sh left under gpurun_out/prof_<round>/ into the tracked summaries under
profiles/: rocprofv3 kernel stats of the sdr:: kernels, PMC traffic per launch (stamped with the hash of the kernel
sources it was measured on), the bench JSON lines, SQ counters and workgroup spans of the FFT kernel.

usage: python tools/make_profiles.py r02            (on the CPU box, after the gpurun call has merged its output)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdrainer_amd.csrc import build  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
key = "c3_f2048"
G = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
P = os.path.join(ROOT, "profiles")


def find(pattern):
    hits = glob.glob(os.path.join(G, pattern), recursive=True)
    assert hits, pattern
    return hits[0]


rows = list(csv.reader(open(find("trace/**/t_kernel_stats.csv"))))
out = [rows[0]] + [r for r in rows[1:] if "sdr::" in r[0]]
csv.writer(open(os.path.join(P, f"{tag}_kernel_stats_{key}.csv"), "w")).writerows(out)


def pmc(name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(find(f"{name}/**/{name}_counter_collection.csv"))):
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
step = sum(v["hbm_bytes_per_launch"] for v in tab.values())
alg = 8 * 2048 * 16384
doc = {
    "_how": ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (no trace domains) over "
             "`python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --serial`; per-dispatch means. Units: "
             "KB. gfx950 correction per MI355X_MICROARCH.md §HBM: FETCH_SIZE reports 1/2 of the bytes of a coalesced "
             "streaming read -> read bytes = FETCH_SIZE*1024*2 (checked here on k_cumulate, which streams exactly the "
             "134.2 MB psd once); WRITE_SIZE is exact."),
    "_source_hash": build.source_hash(),
    "_round": tag,
    key: {"k_fft_psd_hbm_bytes_per_launch": tab[fft]["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": alg,
          "whole_step_hbm_bytes": step, "whole_step_over_algorithmic": round(step / alg, 3), "kernels": tab},
}
json.dump(doc, open(os.path.join(P, "traffic.json"), "w"), indent=1)
for src, dst in (("bench_full.json", f"{tag}_bench_{key}.json"), ("bench_serial.err", f"{tag}_kernel_breakdown_serial.txt"),
                 ("bench_insitu.err", f"{tag}_kernel_breakdown_pipelined.txt"), ("fft_sq_counters.txt", f"{tag}_fft_sq_counters.txt"),
                 ("fft_workgroup_spans.txt", f"{tag}_fft_workgroup_spans.txt"), ("host_input_rate.txt", f"{tag}_host_input_rate.txt"),
                 ("strain_e2e.json", f"{tag}_strain_e2e.json")):
    shutil.copy(os.path.join(G, src), os.path.join(P, dst))
for src, dst in (("ubench_share.txt", f"{tag}_ubench_share.txt"), ("fft_insitu_spans.txt", f"{tag}_fft_insitu_spans.txt"),
                 ("skip_matrix.txt", f"{tag}_skip_matrix.txt")):  # (only there if the diagnostic libraries were built)
    if os.path.exists(os.path.join(G, src)) and os.path.getsize(os.path.join(G, src)) > 0:
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
other = {}
for name in ("bench_insitu", "bench_nodelivery", "bench_graph_c3", "bench_c5", "bench_graph_c5", "bench_c2"):
    d = json.loads(open(os.path.join(G, name + ".json")).read().strip().splitlines()[-1])
    other[name] = {"value_MSamples_per_s": d["value"], "ms_per_step": d["ms_per_step"], "workload": d["config"]["workload"][:40],
                   "frames_per_step": d["config"]["frames_per_step_per_band"], "launch": d["config"].get("launch", "")[:30],
                   "delivery": d["config"].get("delivery", "")[:30]}
json.dump(other, open(os.path.join(P, f"{tag}_other_runs.json"), "w"), indent=1)
for r in out[1:]:
    print(r[0].split("(")[0].replace("void ", "")[:40], "calls", r[1], "avg_ns", r[3])
print("k_fft_psd", json.dumps(tab[fft]))
print("whole step", step, "=", round(step / alg, 3), "x algorithmic")
