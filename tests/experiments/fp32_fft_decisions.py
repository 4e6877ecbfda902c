#!/usr/bin/env python3
"""What would a float32 FFT cost in decisions?  

BASELINE config 3's band (N = 16384, 2 MS/s, 256 keyed carriers) goes through the reference's arithmetic twice:
once as the oracle computes it (complex128 FFT, dsp/fft.go:23-37), once with the FFT alone replaced by a complex64 one
(scipy / pocketfft in single precision) and everything behind it unchanged: PSD rounded to float32, dB projection,
FindNoiseFloor (the oracle's), the two 60-frame float32 rolling means, listen threshold, `value > threshold` per
listener per frame (cw/spectral.go:49), 100-frame cumulation and FindPeaks (the oracle's).

Printed: relative PSD error on noise bins and on carrier bins, flipped keying decisions, changed keying edges, changed
peak lists.  It lives under tests/ because it calls the oracle (test infrastructure); the product never does.

    python tests/experiments/fp32_fft_decisions.py [frames]
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(frames: int = 400, n: int = 16384, rate: int = 2_000_000, tones: int = 256, seed: int = 3003) -> dict:
    import scipy.fft

    from oracle import oracle as orc
    from sdrainer_amd import synth

    edge = synth.default_edge_width(n)
    iq, bins, key = synth.make_band(frames, rate, n, tones, seed=seed, free_last_window=True)
    ref = orc.Receiver(rate, n, edge)
    for b in bins:
        ref.attach(int(b))
    out = ref.process(iq, want_spectrum=True)

    # the float32 path: only the FFT differs
    x = (iq[:, 0::2] + 1j * iq[:, 1::2]).astype(np.complex64)
    X = scipy.fft.fft(x, axis=1)
    assert X.dtype == np.complex64
    X = np.fft.fftshift(X, axes=1)
    psd32 = (X.real.astype(np.float32) ** 2 + X.imag.astype(np.float32) ** 2).astype(np.float32)

    def db(p):  # PSDValueIndB + dBmShift (dsp/fft.go:83-85, rx/receiver.go:376-378); numpy's log10 stands in for Go's
        with np.errstate(divide="ignore"):
            return (10.0 * np.log10(20.0 * p.astype(np.float64) / (float(n) * float(n)))).astype(np.float32) + np.float32(120)

    spec32 = db(psd32)
    ring_nf, ring_dev = np.zeros(60, np.float32), np.zeros(60, np.float32)
    sum_nf = sum_dev = np.float32(0)
    nxt = 0
    raw32 = np.zeros((frames, tones), np.uint8)
    listen_thr32 = np.zeros(frames, np.float32)
    cum = np.zeros(n, np.float32)
    peaks32, count = [], 0
    for f in range(frames):
        mn, var = orc.find_noise_floor(psd32[f], edge)
        dev_in = np.float32(np.float64(db(np.array([np.float32(np.sqrt(var))], np.float32))[0]) * 0.25)
        nf_in = db(np.array([mn], np.float32))[0]
        sum_dev = np.float32(np.float32(sum_dev - ring_dev[nxt]) + dev_in)
        ring_dev[nxt] = dev_in
        sum_nf = np.float32(np.float32(sum_nf - ring_nf[nxt]) + nf_in)
        ring_nf[nxt] = nf_in
        nxt = (nxt + 1) % 60
        noise_dev, noise_floor = np.float32(sum_dev / np.float32(60)), np.float32(sum_nf / np.float32(60))
        thr = np.float32(noise_floor + noise_dev)
        listen_thr32[f] = thr
        raw32[f] = spec32[f, bins] > thr
        cum += spec32[f]
        count += 1
        if count == 100:
            peaks32.append(orc.find_peaks(cum, np.float32(np.float32(15.0) + noise_floor), rate))
            cum[:] = 0
            count = 0

    psd64 = out["psd"]
    carrier = np.zeros(n, bool)
    carrier[bins] = True
    rel = np.abs(psd32.astype(np.float64) - psd64.astype(np.float64)) / np.maximum(psd64.astype(np.float64), 1e-300)
    flips = int(np.count_nonzero(raw32 != out["raw"]))
    edges64 = np.diff(np.concatenate([np.zeros((1, tones), np.int8), out["raw"].astype(np.int8)]), axis=0) != 0
    edges32 = np.diff(np.concatenate([np.zeros((1, tones), np.int8), raw32.astype(np.int8)]), axis=0) != 0
    peak_lists_differ = sum(1 for a, b in zip(peaks32, out["peaks"])
                            if [(p[0], p[1], p[6]) for p in a] != [(p[0], p[1], p[6]) for p in b])
    thr_ulps = np.abs(listen_thr32.view(np.int32).astype(np.int64) - out["frames"]["listen_thr"].view(np.int32).astype(np.int64))
    return {
        "workload": f"config 3 band: N={n}, {tones} carriers, {frames} frames, seed {seed}",
        "psd_rel_error_noise_bins": {"median": float(np.median(rel[:, ~carrier])), "p99": float(np.quantile(rel[:, ~carrier], 0.99)),
                                     "max": float(rel[:, ~carrier].max())},
        "psd_rel_error_carrier_bins_key_down": {"median": float(np.median(rel[:, bins][key.astype(bool)]))},
        "magnitude_tolerance_1e-5_met_on_noise_bins": bool(np.median(rel[:, ~carrier]) / 2 < 1e-5),
        "decisions": int(raw32.size), "decisions_flipped": flips,
        "keying_edges": int(np.count_nonzero(edges64)), "keying_edges_changed": int(np.count_nonzero(edges32 != edges64)),
        "listen_threshold_max_ulps_apart": int(thr_ulps.max()),
        "cumulations": len(peaks32), "peak_lists_that_differ_in_bins": int(peak_lists_differ),
    }


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 400), indent=1))
