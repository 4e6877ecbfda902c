"""Development aid: condense a rocprofv3 --kernel-trace CSV into what a pipeline question needs - per kernel the mean
duration, per queue the busy fraction, the total span, and a compact timeline of a window in the middle.
  python tools/trace_timeline.py <dir or kernel_trace.csv> [--window-ms 3]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    window_ms = float(sys.argv[sys.argv.index("--window-ms") + 1]) if "--window-ms" in sys.argv else 3.0
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], r.get("Queue_Id", "?")))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print(f"{len(rows)} dispatches over {(t1 - t0) / 1e6:.2f} ms")
    per = defaultdict(lambda: [0, 0])
    perq = defaultdict(int)
    for s, e, n, q in rows:
        per[n][0] += 1
        per[n][1] += e - s
        perq[q] += e - s
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:50s} n={c:6d} mean={t / c / 1e3:9.1f} us total={t / 1e6:9.2f} ms")
    for q, t in sorted(perq.items()):
        print(f"  queue {q}: busy {t / (t1 - t0) * 100:.1f} %")
    ours = sorted(r[0] for r in rows if "sdr::" in r[2])
    mid = ours[int(len(ours) * 0.6)] if ours else t0 + (t1 - t0) * 0.6  # inside the timed region, not the start-up
    print(f"timeline from +{(mid - t0) / 1e6:.2f} ms, {window_ms} ms:")
    for s, e, n, q in rows:
        if s >= mid and s < mid + window_ms * 1e6:
            print(f"  {(s - mid) / 1e3:9.1f} .. {(e - mid) / 1e3:9.1f} us  q{q:>3s}  {n}")


if __name__ == "__main__":
    main()
