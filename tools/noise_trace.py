"""Development aid: where the variance chain's consumer (workgroup 7 of k_noise_stats) spends its time.
Needs a library built with -DSDR_NOISE_TRACE (tools/build_abl.sh ntrace "-DSDR_NOISE_TRACE"; SDR_HIP_LIB=...)."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from sdrainer_amd import capi, synth
rate, n, tones, frames = 2_000_000, 16384, 256, 2048
bank = capi.Bank(rate, n, max_batch_frames=frames, max_listeners=tones, max_peaks=1024)
iq, bins, _ = synth.make_band_torch(frames, rate, n, tones, seed=1, device="cuda", free_last_window=True)
for b in bins: bank.attach(0, int(b))
for i in range(4):
    bank.process_device(iq.data_ptr(), frames)
    bank.sync()
out = (C.c_ulonglong * 32)()
L = capi._lib
L.sdr_debug_noise_trace.restype = C.c_int
assert L.sdr_debug_noise_trace(out) == 0
for name, o in (("variance consumer 0 (matrix pipe)", 0), ("window-sum consumer of group 0", 8)):
    total, wait, tiles, spins, clk = out[o], out[o + 1], out[o + 2], out[o + 3], out[o + 4]
    if not total or not tiles:
        print(name + ": no trace")
        continue
    print("%s: total %.1f us, waiting for tiles %.1f us (%.0f%%), %d tiles, %.3f us per tile, %d spins; %.2f GHz, %.1f clocks per term outside the waits"
          % (name, total / 100.0, wait / 100.0, 100.0 * wait / total, tiles, total / 100.0 / tiles, spins, clk / (total * 10.0),
             (clk * (1 - wait / total)) / (tiles * 64.0)))
print("window sums, workgroup 7: chain time / waiting per group (us): " + ", ".join("%.1f / %.1f" % (out[16 + 2 * g] / 100.0, out[17 + 2 * g] / 100.0) for g in range(4)) + "; wave 0 of the workgroup lived %.1f us" % (out[24] / 100.0))
