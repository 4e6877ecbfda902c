"""Development aid: the stage timeline SDR_PROF_TIMELINE wrote (a library built with tools/experiments/prof_timeline.patch
applied - the shipped one does not write it - and run under SDR_BENCH_PROFILE_TIMED=1 python bench.py --steps 20) for one
profiled run - per batch the FFT's start and end and the gaps between FFT launches, then when each stage of the last
batches ended.   python tools/timeline_stages.py <file>"""
import sys

NAMES = ["fft", "scan", "noise_stats", "thresholds", "gather", "cumulate", "find_peaks", "decode"]
rows = [tuple(float(x) for x in l.split()) for l in open(sys.argv[1]) if l.strip() and not l.startswith("#")]
rows = [(int(k), a, b) for k, a, b in rows]
ffts = [(a, b) for k, a, b in rows if k == 0]
print("FFT launches: start, duration, gap to the previous end (ms)")
prev = None
for a, b in ffts:
    print("  %8.3f  %6.3f  %s" % (a, b - a, "" if prev is None else "%6.3f" % (a - prev)))
    prev = b
end = max(b for _, _, b in rows)
print("last FFT ends %.3f, everything ends %.3f (drain %.3f)" % (ffts[-1][1], end, end - ffts[-1][1]))
print("stages ending after the last FFT's start:")
for k, a, b in sorted(rows, key=lambda r: r[1]):
    if b > ffts[-1][0]:
        print("  %-12s %8.3f .. %8.3f (%.3f)" % (NAMES[k] if k < len(NAMES) else str(k), a, b, b - a))
