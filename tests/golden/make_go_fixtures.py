#!/usr/bin/env python3
"""Writes the inputs and the expected digests of integration/go/dump_golden_test.go - the test a maintainer with a Go
toolchain and the reference's modules can run to pin FFT, PSD, dB projection, FindNoiseFloor, thresholds, cumulation and
FindPeaks against the REAL reference, which cannot be built in this image.

Fixtures (integration/go/testdata/):  little-endian float32 IQ, [frames][2 N]
    n512.f32     230 frames of 512 samples, 4 keyed carriers: whole frame loop, two cumulations, FindPeaks
    n4096.f32 / n8192.f32 / n16384.f32   one frame each at the block sizes of BASELINE configs 2, 5 and 3
expected.json: sha256 digests (and the peak lists) the CPU oracle computes for them - the same quantities
tests/golden/dsp_golden.json holds for the longer runs the GPU parity tests use.

    python tests/golden/make_go_fixtures.py
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

OUT = os.path.join(ROOT, "integration", "go", "testdata")
CASES = [  # name, sample rate, block size, tones, frames, seed, free_last_window  (seeds of make_dsp_golden.py)
    ("n512", 48000, 512, 4, 230, 4101, False),
    ("n4096", 192000, 4096, 16, 1, 4102, False),
    ("n8192", 2000000, 8192, 16, 1, 4103, True),
    ("n16384", 2000000, 16384, 32, 1, 4104, True),
]
CENTER = 7020000


def digest(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    os.makedirs(OUT, exist_ok=True)
    expected = {}
    for name, rate, n, tones, frames, seed, free_last in CASES:
        iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=seed, free_last_window=free_last)
        iq = np.ascontiguousarray(iq, dtype="<f4")
        iq.tofile(os.path.join(OUT, name + ".f32"))
        edge = synth.default_edge_width(n)
        ref = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=CENTER)
        out = ref.process(iq, want_spectrum=True)
        recs = out["frames"]
        expected[name] = {
            "file": name + ".f32", "sample_rate": rate, "block_size": n, "frames": frames, "edge_width": edge,
            "center_frequency": CENTER,
            "spectrum_sha256": digest(out["spectrum"]), "psd_sha256": digest(out["psd"]),
            "records_sha256": {f: digest(recs[f]) for f in recs.dtype.names if f != "pad"},
            "cumulation_sha256": [digest(out["cumulation"][c]) for c in range(out["n_chunks"])],
            "peaks": [[list(p) for p in out["peaks"][c]] for c in range(out["n_chunks"])],
        }
        print(name, iq.nbytes, "bytes,", out["n_chunks"], "cumulations")
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump(expected, f, indent=1)


if __name__ == "__main__":
    main()
