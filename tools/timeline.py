#!/usr/bin/env python3
"""Development aid: condenses a rocprofv3 --kernel-trace CSV of a pipelined bench run into a per-kernel
table (average duration, share of wall time, average number of other kernels running beside it) and the
critical-path view per batch (when each kernel of a steady-state batch starts and ends relative to its FFT)."""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("sdr::", "").replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2  # steady state: second half
rows = rows[skip:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
wall = (t1 - t0) / 1e3
agg = defaultdict(lambda: [0, 0.0])
for s, e, n in rows:
    agg[n][0] += 1
    agg[n][1] += (e - s) / 1e3
n_fft = agg.get("k_fft_psd", [1])[0]
print(f"window {wall:.1f} us, {n_fft} batches -> {wall / max(n_fft, 1):.1f} us per batch")
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:18s} {c:5d} launches  avg {us / c:8.1f} us   busy {100 * us / wall:5.1f} % of wall")
# per-batch relative schedule: group by occurrence index
byname = defaultdict(list)
for s, e, n in rows:
    byname[n].append((s, e))
k = min(len(v) for v in byname.values()) - 2
print("steady-state schedule of one batch (us relative to its FFT start; averaged):")
ffts = byname["k_fft_psd"]
for n, v in byname.items():
    # align the j-th launch of each kernel to the j-th FFT (kernels are launched once per batch)
    off = len(v) - len(ffts)
    ds, de = [], []
    for j in range(2, k):
        jf = j
        jv = j + (off if off < 0 else 0)
        if 0 <= jv < len(v) and jf < len(ffts):
            ds.append((v[jv][0] - ffts[jf][0]) / 1e3)
            de.append((v[jv][1] - ffts[jf][0]) / 1e3)
    if ds:
        print(f"  {n:18s} start {sum(ds) / len(ds):8.1f}  end {sum(de) / len(de):8.1f}")
