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
out = (C.c_ulonglong * 8)()
L = capi._lib
L.sdr_debug_noise_trace.restype = C.c_int
assert L.sdr_debug_noise_trace(out) == 0
total, wait, tiles, spins = out[0], out[1], out[2], out[3]
print("consumer: total %d ticks (100 MHz) = %.1f us, waiting for tiles %.1f us (%.0f%%), %d tiles, %.3f us per tile, %d spins"
      % (total, total / 100.0, wait / 100.0, 100.0 * wait / max(total, 1), tiles, total / 100.0 / max(tiles, 1), spins))
print("shader clock during the chain: %.2f GHz; %.1f clocks per term outside the waits" % (out[4] / (total * 10.0), (out[4] * (1 - wait / total)) / (tiles * 64.0)))
