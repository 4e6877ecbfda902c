"""Development aid: host time to enqueue one step (all launches, events, waits) against the step itself."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from sdrainer_amd import capi, synth
rate, n, tones, frames = 2_000_000, 16384, 256, 2048
bank = capi.Bank(rate, n, max_batch_frames=frames, max_listeners=tones, max_peaks=1024)
bank.set_stream(torch.cuda.current_stream().cuda_stream)
iq, bins, _ = synth.make_band_torch(frames, rate, n, tones, seed=1, device="cuda", free_last_window=True)
for b in bins: bank.attach(0, int(b))
for mode in ("plain", "results+poll"):
    if mode != "plain":
        bank.enable_results(True)
    for i in range(5): bank.process_device(iq.data_ptr(), frames)
    torch.cuda.synchronize()
    while mode != "plain" and bank.poll_counts() is not None: pass
    for steps in (4, 40):
        t0 = time.perf_counter()
        for i in range(steps):
            bank.process_device(iq.data_ptr(), frames)
            if mode != "plain":
                while bank.poll_counts() is not None: pass
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        while mode != "plain" and bank.poll_counts(wait=True) is not None: pass
        print("%-13s %2d steps: enqueue per step %.1f us ; total per step %.1f us" % (mode, steps, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
