import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sdr::", "").replace("r32::", "")[:22]
    if "sdr::" in r["Kernel_Name"]:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
ffts = [r for r in rows if "k_fft" in r[2]]
# the last 20 FFT launches = the timed region (warmup 5 before, settle before that)
timed = ffts[-20:]
t0 = timed[0][0]
print("FFT launches of the timed region: start, duration, gap to the previous end (us)")
prev = None
for s, e, n in timed:
    print(f"  {(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:7.1f}  {'' if prev is None else f'{(s - prev) / 1e3:6.1f}'}")
    prev = e
last_fft_end = timed[-1][1]
print(f"last FFT ends at {(last_fft_end - t0) / 1e3:.1f} us; kernels after it:")
for s, e, n in rows:
    if e > last_fft_end - 50_000 and s >= t0:
        if s > last_fft_end - 700_000:
            print(f"  {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:6.1f})  {n}")
print(f"last kernel ends at {(max(r[1] for r in rows) - t0) / 1e3:.1f} us")
