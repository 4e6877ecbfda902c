"""Development aid: the FFT workgroups' spans INSIDE the running pipeline (library built with
-DSDR_FFT_CLOCK -DSDR_FFT_CLOCK_LIB).  Prints the distribution of workgroup lifetimes and, per CU, how long the CU
went without an FFT workgroup between two of them.  usage: SDR_HIP_LIB=tools/abl/libfftclk.so python tools/insitu_fft.py [alone]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from sdrainer_amd import capi, synth
rate, n, tones, frames = 2_000_000, 16384, 256, 2048
bank = capi.Bank(rate, n, max_batch_frames=frames, max_listeners=tones, max_peaks=1024)
bank.set_stream(torch.cuda.current_stream().cuda_stream)
ring = []
for seed in (1, 2, 3):
    iq, bins, _ = synth.make_band_torch(frames, rate, n, tones, seed=seed, device="cuda", free_last_window=True)
    ring.append(iq)
for b in bins: bank.attach(0, int(b))
L = capi._lib
L.sdr_debug_fft_wg.restype = C.c_int
def run(steps):
    for i in range(steps):
        bank.process_device(ring[i % 3].data_ptr(), frames)
run(60)
torch.cuda.synchronize()
out = np.zeros((2048, 4), np.uint64)
samples = []
for rep in range(6):
    run(40)
    time.sleep(0.004)  # the host is far ahead: this reads in the middle of the queued work
    assert L.sdr_debug_fft_wg(out.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
    samples.append(out.copy())
    torch.cuda.synchronize()
for k, wg in enumerate(samples):
    start, end = wg[:, 0].astype(np.int64), wg[:, 1].astype(np.int64)
    dur = (end - start) / 100.0  # 100 MHz -> us
    # entries come from two consecutive launches (the later launch has overwritten the first workgroups): split at the jump
    order = np.argsort(start)
    s_sorted = start[order]
    jump = np.argmax(np.diff(s_sorted)) if len(s_sorted) > 1 else 0
    hw = wg[:, 3]
    cu = ((hw >> np.uint64(32)) & np.uint64(15)).astype(np.int64) * 1000 + ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64) * 100 + ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
    print("sample %d: lifetime us: median %.1f  p10 %.1f  p90 %.1f  max %.1f ; launches span %.1f us ; largest start gap %.1f us"
          % (k, np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), dur.max(), (end.max() - start.min()) / 100.0,
             np.diff(s_sorted).max() / 100.0))
    # per CU: idle time between consecutive FFT workgroups (within the newer launch only)
    newer = order[jump + 1:] if np.diff(s_sorted).max() > 3000 else order
    gaps, counts = [], []
    for c in np.unique(cu[newer]):
        idx = newer[cu[newer] == c]
        idx = idx[np.argsort(start[idx])]
        counts.append(len(idx))
        g = (start[idx][1:] - end[idx][:-1]) / 100.0
        gaps.extend(g.tolist())
    gaps = np.array(gaps) if gaps else np.zeros(1)
    print("          newer launch: %d workgroups on %d CUs (%.1f per CU: min %d max %d); gap between a CU's consecutive FFT workgroups: median %.1f us  p90 %.1f  max %.1f  sum/CU %.1f us"
          % (len(newer), len(counts), np.mean(counts), min(counts), max(counts), np.median(gaps), np.percentile(gaps, 90), gaps.max(), gaps.sum() / max(len(counts), 1)))
