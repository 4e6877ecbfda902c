import sys, numpy as np, torch
sys.path.insert(0, '.')
from sdrainer_amd import capi, synth
n, rate, tones, frames = 16384, 2_000_000, 256, 8192
edge = synth.default_edge_width(n)
bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=frames, max_listeners=tones, max_peaks=512)
x = synth.make_band_torch(frames, rate, n, tones, 3000, torch.device('cuda'), free_last_window=True)
iq = x[0] if isinstance(x, tuple) else x
for rep in range(3):
    bank.process_device(iq.data_ptr(), frames)
    bank.sync()
    import ctypes as C
    rec = np.zeros(frames, capi.FRAME_REC_DTYPE)
    # raw copy without the exact check: read through the library (the check replaces variance only)
    r = bank.read_frame_records(0)
    pad = r["pad"]
    vals, cnt = np.unique(pad, return_counts=True)
    print("batch", rep, dict(zip(vals.tolist(), cnt.tolist())))
    idx = np.flatnonzero(pad != 0)[:10]
    print("  first flagged frames", idx.tolist(), "min_mean", r["min_mean"][idx].tolist()[:3], "variance", r["variance"][idx].tolist()[:3])
