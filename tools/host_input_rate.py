"""Development aid: the PCIe-inclusive rate of the host-buffer boundary (sdr_push_iq + sdr_process_staged),
config 3 geometry.  Never the bench's `value` (that one has its input resident in HBM); DESIGN.md quotes it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sdrainer_amd import capi, synth

rate, n, tones, frames = 2_000_000, 16384, 256, 2048
edge = synth.default_edge_width(n)
bank = capi.Bank(rate, n, n_bands=1, edge_width=edge, peak_threshold=15.0, signal_debounce=1, max_listeners=tones,
                 max_batch_frames=frames, max_peaks=1024, find_peaks=True, trace=False, device_id=0)
iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=3017, free_last_window=True)
for b in bins:
    bank.attach(0, int(b))
for kind in ("float32", "kiwi"):
    if kind == "kiwi":
        per = n * 2
        q = np.clip(iq * 32767.0 * 8, -32768, 32767).astype(">i2")  # what a KiwiSDR sends: big-endian int16
        # (a payload may hold several frames: kiwi/kiwi.go:94-105 splits it; 256 frames per message here so that the
        # Python call overhead does not bound the measurement)
        payloads = [bytes(17) + q[f:f + 256].tobytes() for f in range(0, frames, 256)]
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 12
        for _ in range(steps):
            if kind == "float32":
                assert bank.push_iq(0, rate, iq) == 0
            else:
                for p in payloads:
                    assert bank.push_kiwi_snd(0, rate, p) == 0
            assert bank.process_staged() == frames
        bank.sync()
        dt = time.perf_counter() - t0
        print(f"{kind:8s} host input: {steps * frames * n / dt / 1e6:9.1f} MSamples/s  ({dt / steps * 1e3:.2f} ms per 2048-frame batch, "
              f"{steps * frames * n * (8 if kind == 'float32' else 4) / dt / 1e9:.1f} GB/s over the boundary)")
bank.close()
