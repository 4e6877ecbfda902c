#!/usr/bin/env python3
"""What would ALLOWING FMA in the FFT change?  (VERDICT r03 task 6)

The reference's arithmetic never fuses a multiply-add (Go on amd64), and the shipped kernel reproduces that: ten float64
instructions per butterfly.  A contracting compiler needs eight (the complex product as a multiply and a fused
multiply-add per component) - measured on the GPU by tools/fft_bench built with -ffp-contract=fast.  This script
measures the OTHER side: BASELINE config 3's band goes through the reference's arithmetic twice - once as the oracle
computes it, once with only the FFT and the psd contracted (tests/experiments/fma_fft.c) and everything behind them
unchanged (FindNoiseFloor, the two float32 rolling means, the listen threshold, `value > threshold` per listener per
frame, the 100-frame cumulation and FindPeaks: the oracle's) - and counts what differs: psd words, keying decisions,
keying edges, peak lists.  Test infrastructure (it calls the oracle); the product never does.

    python tests/experiments/fma_fft_decisions.py [frames]        (8192 frames: a few minutes)
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _lib():
    so = os.path.join(tempfile.gettempdir(), "libfma_fft_experiment.so")
    src = os.path.join(ROOT, "tests", "experiments", "fma_fft.c")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "-fvisibility=hidden", "-o", so, src, "-lm"])
    L = C.CDLL(so)
    L.fma_iq_to_psd.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def run(frames: int = 400, n: int = 16384, rate: int = 2_000_000, tones: int = 256, seed: int = 3003, chunk: int = 512) -> dict:
    from oracle import oracle as orc
    from sdrainer_amd import synth

    L = _lib()
    _db = orc.lib().orc_psd_value_in_db

    def psd_db(v):  # dsp.PSDValueIndB, the oracle's own logarithm
        return np.float32(_db(C.c_float(float(v)), n))

    w = orc.radix2_factors(n)
    wre, wim = np.ascontiguousarray(w.real), np.ascontiguousarray(w.imag)
    edge = synth.default_edge_width(n)
    iq_all, bins, _ = synth.make_band(frames, rate, n, tones, seed=seed, free_last_window=True)
    ref = orc.Receiver(rate, n, edge)
    for b in bins:
        ref.attach(int(b))

    ring_nf, ring_dev = np.zeros(60, np.float32), np.zeros(60, np.float32)
    sum_nf = sum_dev = np.float32(0)
    nxt = 0
    cum = np.zeros(n, np.float32)
    count = 0
    words = differing = 0
    max_ulp = 0
    flips = decisions = edges_ref = edges_changed = 0
    peak_lists = peak_lists_differ = 0
    last_ref = np.zeros(tones, np.uint8)
    last_fma = np.zeros(tones, np.uint8)
    thr_ulps_max = 0
    psd = np.empty(n, np.float32)
    for c0 in range(0, frames, chunk):
        iq = iq_all[c0:c0 + chunk]
        out = ref.process(iq, want_spectrum=True)
        k_peak = 0
        for f in range(iq.shape[0]):
            L.fma_iq_to_psd(n, iq[f].ctypes.data, wre.ctypes.data, wim.ctypes.data, psd.ctypes.data)
            a, b = psd.view(np.int32).astype(np.int64), out["psd"][f].view(np.int32).astype(np.int64)
            d = np.abs(a - b)
            words += n
            differing += int(np.count_nonzero(d))
            max_ulp = max(max_ulp, int(d.max()))
            mn, var = orc.find_noise_floor(psd, edge)
            dev_in = np.float32(np.float64(psd_db(np.float32(np.sqrt(var))) + np.float32(120)) * 0.25)
            nf_in = psd_db(mn) + np.float32(120)
            sum_dev = np.float32(np.float32(sum_dev - ring_dev[nxt]) + dev_in)
            ring_dev[nxt] = dev_in
            sum_nf = np.float32(np.float32(sum_nf - ring_nf[nxt]) + nf_in)
            ring_nf[nxt] = nf_in
            nxt = (nxt + 1) % 60
            noise_dev, noise_floor = np.float32(sum_dev / np.float32(60)), np.float32(sum_nf / np.float32(60))
            thr = np.float32(noise_floor + noise_dev)
            thr_ulps_max = max(thr_ulps_max, abs(int(thr.view(np.int32)) - int(out["frames"]["listen_thr"][f].view(np.int32))))
            spec_bins = np.array([psd_db(v) for v in psd[bins]], np.float32) + np.float32(120)
            raw = (spec_bins > thr).astype(np.uint8)
            ref_raw = out["raw"][f]
            decisions += tones
            flips += int(np.count_nonzero(raw != ref_raw))
            e_ref, e_fma = ref_raw != last_ref, raw != last_fma
            edges_ref += int(np.count_nonzero(e_ref))
            edges_changed += int(np.count_nonzero(e_ref != e_fma))
            last_ref, last_fma = ref_raw.copy(), raw
            # the cumulation wants the dB of every bin: numpy's log10 is within an ulp of the oracle's, and a last-bit
            # difference of the float32 dB cannot be told from the FFT's own; the oracle's logarithm on the bins that matter
            with np.errstate(divide="ignore"):
                spec = (10.0 * np.log10(20.0 * psd.astype(np.float64) / (float(n) * float(n)))).astype(np.float32) + np.float32(120)
            cum += spec
            count += 1
            if count == 100:
                pk = orc.find_peaks(cum, np.float32(np.float32(15.0) + noise_floor), rate)
                want = out["peaks"][k_peak]
                k_peak += 1
                peak_lists += 1
                if [(p[0], p[1], p[6]) for p in pk] != [(p[0], p[1], p[6]) for p in want]:
                    peak_lists_differ += 1
                cum[:] = 0
                count = 0
    return {
        "workload": f"config 3 band: N={n}, {tones} carriers, {frames} frames, seed {seed}",
        "psd_words": words, "psd_words_that_differ": differing, "fraction": differing / max(words, 1), "max_ulps_apart": max_ulp,
        "decisions": decisions, "decisions_flipped": flips, "keying_edges": edges_ref, "keying_edges_changed": edges_changed,
        "listen_threshold_max_ulps_apart": thr_ulps_max, "cumulations": peak_lists, "peak_lists_that_differ_in_bins": peak_lists_differ,
    }


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 400), indent=1))
