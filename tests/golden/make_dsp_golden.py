#!/usr/bin/env python3
"""Generates tests/golden/dsp_golden.json: digests of what the CPU oracle (oracle/sdr_oracle.c) computes for
seeded synthetic bands, for the rows of SURVEY.md 8(a) the reference's own tests do not pin (FFT, projection,
noise floor, thresholds, cumulation, FindPeaks, the frame loop).

The reference (Go) cannot be run here, so these are the ORACLE's outputs, not the reference's: they pin the
oracle against regressions and let the GPU path be checked against committed data.  Inputs are regenerated
from the seed (sdrainer_amd.synth.make_band: numpy PCG64 + exact-bin tones), so the fixture holds digests and
a few spot values only.

    python tests/golden/make_dsp_golden.py            # rewrites dsp_golden.json
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from sdrainer_amd import synth  # noqa: E402

CASES = [  # name, sample rate, block size, tones, frames, seed, free_last_window
    ("n512", 48000, 512, 4, 230, 4101, False),
    ("n4096", 192000, 4096, 16, 120, 4102, False),
    ("n8192", 2000000, 8192, 16, 110, 4103, True),
    ("n16384", 2000000, 16384, 32, 105, 4104, True),
]


def digest(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(name, rate, n, tones, frames, seed, free_last):
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=seed, free_last_window=free_last)
    edge = synth.default_edge_width(n)
    ref = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=7020000)
    lids = [ref.attach(int(b)) for b in bins]
    out = ref.process(iq, want_spectrum=True)
    recs = out["frames"]
    case = {
        "sample_rate": rate, "block_size": n, "tones": tones, "frames": frames, "seed": seed,
        "free_last_window": free_last, "edge_width": edge, "bins": [int(b) for b in bins],
        "iq_sha256": digest(iq),
        "spectrum_sha256": digest(out["spectrum"]), "psd_sha256": digest(out["psd"]),
        "records_sha256": {f: digest(recs[f]) for f in recs.dtype.names if f != "pad"},
        "keying_sha256": digest(out["deb"]), "values_sha256": digest(out["values"]),
        "cumulation_sha256": [digest(out["cumulation"][c]) for c in range(out["n_chunks"])],
        "peaks": [[list(p) for p in out["peaks"][c]] for c in range(out["n_chunks"])],
        "text": [ref.text(l) for l in lids],
        # spot values, readable in a diff
        "spot": {"spectrum[0][n/2]": float(out["spectrum"][0][n // 2]), "psd[0][bin0]": float(out["psd"][0][int(bins[0])]),
                 "noise_floor[last]": float(recs["noise_floor"][-1]), "listen_thr[last]": float(recs["listen_thr"][-1])},
    }
    return case


def main():
    doc = {"_generator": "tests/golden/make_dsp_golden.py", "_source": "CPU oracle (oracle/sdr_oracle.c), not the Go reference",
           "cases": {c[0]: run_case(*c) for c in CASES}}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dsp_golden.json"), "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in doc["cases"].items():
        print(k, v["spectrum_sha256"][:16], len(v["peaks"]), "cumulations", [len(p) for p in v["peaks"]], v["text"][:2])


if __name__ == "__main__":
    main()
