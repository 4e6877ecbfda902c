#!/usr/bin/env python3
"""Condenses what tools/run_profiles.sh left under gpurun_out/prof_<round>/ into the tracked summaries under
profiles/: rocprofv3 kernel stats of the sdr:: kernels, PMC traffic per launch for configs 3 and 5 (stamped with the
hash of the kernel sources it was measured on), the bench JSON lines, and the standalone FFT evidence (production
binary, per-workgroup clock, ablation matrix, SQ counters) together with the manifest of the binaries that produced it.

usage: python tools/make_profiles.py r03            (on the CPU box, after the gpurun call has merged its output)
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

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
key = "c3_f8192"  # bench.py's default batch for config 3
G = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
P = os.path.join(ROOT, "profiles")


def find(pattern):
    hits = glob.glob(os.path.join(G, pattern), recursive=True)
    assert hits, pattern
    return hits[0]


def last_json(name):
    return json.loads(open(os.path.join(G, name)).read().strip().splitlines()[-1])


measured_hash = open(os.path.join(G, "README.txt")).read().split()[-1]
if measured_hash != build.source_hash():
    print(f"NOTE: measured on sources {measured_hash[:12]}, the tree is now {build.source_hash()[:12]}")

rows = list(csv.reader(open(find("trace/**/t_kernel_stats.csv"))))
out = [rows[0]] + [r for r in rows[1:] if "sdr::" in r[0]]
csv.writer(open(os.path.join(P, f"{tag}_kernel_stats_{key}.csv"), "w")).writerows(out)
try:
    rows5 = list(csv.reader(open(find("trace_c5/**/t_kernel_stats.csv"))))
    csv.writer(open(os.path.join(P, f"{tag}_kernel_stats_c5_f2048.csv"), "w")).writerows([rows5[0]] + [r for r in rows5[1:] if "sdr::" in r[0]])
except AssertionError:
    print("missing: config-5 kernel trace")


def pmc(directory, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(find(f"{directory}/**/{name}_counter_collection.csv"))):
        k = r["Kernel_Name"]
        # (k_noise_exact_check belongs to the record read behind the run - sdr_read_frame_records - not to a step)
        if "sdr::" in k and "k_noise_exact_check" not in k and "k_mfma_order_probe" not in k:
            agg[k.split("(")[0].replace("void ", "").replace("sdr::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def traffic(fetch_dir, write_dir, alg):
    f, w = pmc(fetch_dir, "fetch"), pmc(write_dir, "write")
    tab = {}
    for k in f:
        tab[k] = {"FETCH_SIZE_KB_raw": round(f[k], 1), "WRITE_SIZE_KB_raw": round(w.get(k, 0), 1),
                  "hbm_read_bytes_corrected": int(f[k] * 1024 * 2), "hbm_write_bytes": int(w.get(k, 0) * 1024),
                  "hbm_bytes_per_launch": int(f[k] * 1024 * 2 + w.get(k, 0) * 1024)}
    fft = [k for k in tab if "k_fft" in k][0]  # (k_fft_psd<..> or r32::k_fft_r32)
    step = sum(v["hbm_bytes_per_launch"] for v in tab.values())
    return {"k_fft_psd_hbm_bytes_per_launch": tab[fft]["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": alg,
            "whole_step_hbm_bytes": step, "whole_step_over_algorithmic": round(step / alg, 3),
            "whole_step_bytes_per_sample": round(step / (alg / 8), 2), "kernels": tab}


doc = {
    "_how": ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (no trace domains) over "
             "`python3 bench.py [--workload c5] --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --serial`; per-dispatch "
             "means. Units: KB. gfx950 correction per MI355X_MICROARCH.md §HBM: FETCH_SIZE reports 1/2 of the bytes of a "
             "coalesced streaming read -> read bytes = FETCH_SIZE*1024*2 (checked on k_cumulate, which streams exactly "
             "the psd once); WRITE_SIZE is exact.  algorithmic = 8 B per complex64 sample in + 0 out (the psd is an "
             "intermediate of the path, SURVEY 8(d))"),
    "_source_hash": measured_hash,
    "_round": tag,
    key: traffic("fetch", "write", 8 * 8192 * 16384),
    "c5_f2048": traffic("fetch_c5", "write_c5", 8 * 8 * 2048 * 8192),
}
json.dump(doc, open(os.path.join(P, "traffic.json"), "w"), indent=1)

copies = [("bench_full.json", f"{tag}_bench_{key}.json"), ("bench_serial.err", f"{tag}_kernel_breakdown_serial.txt"),
          ("bench_insitu.err", f"{tag}_kernel_breakdown_pipelined.txt"), ("fft_sq_counters.txt", f"{tag}_fft_sq_counters.txt"),
          ("fft_standalone.txt", f"{tag}_fft_standalone.txt"), ("fft_standalone_f8192.txt", f"{tag}_fft_standalone_f8192.txt"), ("fft_ablation_summary.txt", f"{tag}_fft_ablation_matrix.txt"),
          ("fft_phases.txt", f"{tag}_fft_phase_order.txt"), ("tool_manifest.txt", f"{tag}_tool_manifest.txt"),
          ("host_input_rate.txt", f"{tag}_host_input_rate.txt"), ("strain_e2e.json", f"{tag}_strain_e2e.json"),
          ("mfma_f64.txt", f"{tag}_mfma_f64_probe.txt"), ("cu_time.txt", f"{tag}_cu_time_per_kernel.txt"),
          ("noise_trace.txt", f"{tag}_noise_consumer_trace.txt"), ("decode_clocks.txt", f"{tag}_decode_stage_clocks.txt"),
          ("ubench_f64.txt", f"{tag}_ubench_f64.txt"), ("ubench_cvt.txt", f"{tag}_ubench_cvt.txt"), ("r32_standalone.txt", f"{tag}_fft_r32_standalone.txt"),
          ("r32_phases.txt", f"{tag}_fft_r32_phase_order.txt"), ("fences.txt", f"{tag}_fences_priced.txt"),
          ("noise_paths.txt", f"{tag}_noise_scan_vs_chains.txt"), ("box_probe.txt", f"{tag}_box_probe.txt")]
for src, dst in copies:
    s = os.path.join(G, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copy(s, os.path.join(P, dst))
    else:
        print("missing:", src)


def line(name):
    d = last_json(name + ".json")
    return {"value_MSamples_per_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "workload": d["config"]["workload"][:40],
            "frames_per_step": d["config"]["frames_per_step_per_band"], "launch": d["config"].get("launch", "")[:30],
            "delivery": d["config"].get("delivery", "")[:30], "fft_avg_ms": d.get("roofline", {}).get("avg_launch_ms")}


other = {}
names = ["bench_steps20", "bench_steps20_b", "bench_steps20_c", "bench_f2048", "bench_f2048_steps20", "bench_f4096", "bench_insitu", "bench_nodelivery", "bench_c5",
         "bench_c2", "bench_c2_f8192", "bench_c3_r32off", "bench_c3_chains", "bench_c5_chains"] + \
        [f"bench_graph_c5_{i}" for i in range(1, 6)] + [f"bench_graph_c3_{i}" for i in range(1, 4)]
for name in names:
    try:
        other[name] = line(name)
    except Exception as e:  # a run that failed stays visible
        other[name] = {"error": str(e)[:200]}
json.dump(other, open(os.path.join(P, f"{tag}_other_runs.json"), "w"), indent=1)
for r in out[1:]:
    print(r[0].split("(")[0].replace("void ", "")[:40], "calls", r[1], "avg_ns", r[3])
for k in (key, "c5_f2048"):
    print(k, "whole step", doc[k]["whole_step_hbm_bytes"], "=", doc[k]["whole_step_over_algorithmic"], "x algorithmic,",
          doc[k]["whole_step_bytes_per_sample"], "B/sample")
