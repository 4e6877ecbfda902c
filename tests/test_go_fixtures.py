"""integration/go: the committed IQ fixtures and expected digests of dump_golden_test.go (the test a maintainer with a Go
toolchain runs to pin FFT / PSD / dB / FindNoiseFloor / thresholds / cumulation / FindPeaks against the real reference).
Go is not in this image, so what CAN be checked here is: the fixtures are what the generator writes, and expected.json is
what the oracle computes for exactly those bytes."""
import hashlib
import json
import os

import numpy as np

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "integration", "go", "testdata")


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_expected_json_is_the_oracle_on_the_committed_fixtures():
    expected = json.load(open(os.path.join(DATA, "expected.json")))
    assert sorted(expected) == ["n16384", "n4096", "n512", "n8192"]
    for name, c in expected.items():
        n, frames = c["block_size"], c["frames"]
        iq = np.fromfile(os.path.join(DATA, c["file"]), dtype="<f4").reshape(frames, 2 * n)
        ref = orc.Receiver(c["sample_rate"], n, c["edge_width"], 15.0, 1, center_frequency=c["center_frequency"])
        out = ref.process(iq, want_spectrum=True)
        assert _digest(out["psd"]) == c["psd_sha256"] and _digest(out["spectrum"]) == c["spectrum_sha256"], name
        for f, d in c["records_sha256"].items():
            assert _digest(out["frames"][f]) == d, (name, f)
        assert [_digest(out["cumulation"][k]) for k in range(out["n_chunks"])] == c["cumulation_sha256"]
        assert [[list(p) for p in out["peaks"][k]] for k in range(out["n_chunks"])] == c["peaks"]
    # the whole frame loop is in the n512 case: two completed cumulations, peaks found
    assert len(expected["n512"]["peaks"]) == 2 and len(expected["n512"]["peaks"][1]) > 0


def test_go_test_says_what_it_is():
    src = open(os.path.join(ROOT, "integration", "go", "dump_golden_test.go")).read()
    assert "HAS NEVER BEEN COMPILED" in src and "dsp.FindNoiseFloor" in src and "dsp.FindPeaks" in src
