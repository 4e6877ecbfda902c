"""Committed digests of the unpinned rows (tests/golden/dsp_golden.json, made by tests/golden/make_dsp_golden.py
from the CPU oracle): the oracle must keep reproducing them (CPU), and the HIP path must produce them (GPU)."""
import hashlib
import json
import os

import numpy as np
import pytest

from sdrainer_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "dsp_golden.json")) as f:
    GOLDEN = json.load(f)["cases"]


def digest(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_input(c):
    iq, bins, _ = synth.make_band(c["frames"], c["sample_rate"], c["block_size"], c["tones"], seed=c["seed"],
                                  free_last_window=c["free_last_window"])
    assert digest(iq) == c["iq_sha256"], "the synthetic generator no longer reproduces the fixture's input"
    assert [int(b) for b in bins] == c["bins"]
    return iq, bins


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_oracle_reproduces_golden(name):
    from oracle import oracle as orc
    c = GOLDEN[name]
    iq, bins = make_input(c)
    ref = orc.Receiver(c["sample_rate"], c["block_size"], c["edge_width"], 15.0, 1, center_frequency=7020000)
    lids = [ref.attach(int(b)) for b in bins]
    out = ref.process(iq, want_spectrum=True)
    assert digest(out["spectrum"]) == c["spectrum_sha256"]
    assert digest(out["psd"]) == c["psd_sha256"]
    for f, h in c["records_sha256"].items():
        assert digest(out["frames"][f]) == h, f
    assert digest(out["deb"]) == c["keying_sha256"] and digest(out["values"]) == c["values_sha256"]
    assert [digest(out["cumulation"][k]) for k in range(out["n_chunks"])] == c["cumulation_sha256"]
    assert [[list(p) for p in out["peaks"][k]] for k in range(out["n_chunks"])] == c["peaks"]
    assert [ref.text(l) for l in lids] == c["text"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_gpu_reproduces_golden(name):
    from sdrainer_amd import capi
    c = GOLDEN[name]
    iq, bins = make_input(c)
    n, frames = c["block_size"], c["frames"]
    bank = capi.Bank(c["sample_rate"], n, edge_width=c["edge_width"], max_batch_frames=256, max_listeners=c["tones"],
                     trace=True, max_peaks=512)
    bank.set_center_frequency(0, 7020000)
    lids = [bank.attach(0, int(b)) for b in bins]
    assert bank.process_host(iq) == frames
    sp = np.empty((frames, n), np.float32)
    pd = np.empty((frames, n), np.float32)
    for f in range(frames):
        sp[f], pd[f] = bank.read_spectrum(0, f)
    assert digest(sp) == c["spectrum_sha256"] and digest(pd) == c["psd_sha256"]
    recs = bank.read_frame_records(0)
    for f, h in c["records_sha256"].items():
        assert digest(np.ascontiguousarray(recs[f])) == h, f
    deb = np.stack([bank.read_keying_bits(0, l) for l in lids], axis=1).astype(np.uint8)
    vals = np.stack([bank.read_trace(0, l)[0] for l in lids], axis=1).astype(np.float32)
    assert digest(deb) == c["keying_sha256"] and digest(vals) == c["values_sha256"]
    assert bank.last_batch_chunks == len(c["peaks"])
    for k in range(len(c["peaks"])):
        assert digest(bank.read_cumulation(0, k)) == c["cumulation_sha256"][k]
        peaks, count, _ = bank.read_peaks(0, k)
        assert [list(p) for p in peaks] == c["peaks"][k]
    assert [bank.read_text(0, l) for l in lids] == c["text"]
    bank.close()
