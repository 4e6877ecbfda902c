"""Randomised parity: the HIP path against the oracle on streams whose block size, length, number and strength of
signals, debounce threshold and cuts into batches (down to single frames) are drawn from a seed - frame records, keying
bits, edges, decoder state, text and the peaks of every completed cumulation, bit for bit.  Complements the fixed cases of
test_gpu_parity.py: the staged decoder, the cumulation's bound-and-refine path and the carried state all see shapes nobody
picked by hand."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from sdrainer_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from sdrainer_amd import capi as c
    c.load()
    return c


def _bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("seed", range(int(os.environ.get("SDR_FUZZ_SEEDS", "10"))))  # (a longer soak: SDR_FUZZ_SEEDS=300)
def test_random_streams_and_cuts(capi, seed):
    rng = np.random.default_rng(4242 + seed)
    n = int(rng.choice([512, 1024, 2048]))
    if os.environ.get("SDR_FUZZ_N"):  # (tests/test_forced_paths.py: N = 16384 through k_fft_r32 and its wide tap)
        n = int(os.environ["SDR_FUZZ_N"])
    rate = {512: 48000, 1024: 96000, 2048: 192000}.get(n, 2_000_000)
    frames = int(rng.integers(230, 620 if n <= 2048 else 420))
    tones = int(rng.integers(2, 11))
    debounce = int(rng.choice([1, 1, 2, 3, 5]))
    weak = bool(rng.integers(0, 2))
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=9000 + seed,
                                  amplitude=synth.TONE_AMPLITUDE * (0.0027 if weak else 1.0))
    edge = synth.default_edge_width(n)
    ref = orc.Receiver(rate, n, edge, 15.0, debounce, center_frequency=14_030_000)
    bank = capi.Bank(rate, n, edge_width=edge, signal_debounce=debounce, max_batch_frames=256, max_listeners=tones + 1,
                     trace=True, max_peaks=256)
    bank.set_center_frequency(0, 14_030_000)
    lids = [bank.attach(0, int(b)) for b in bins]
    assert lids == [ref.attach(int(b)) for b in bins]
    out = ref.process(iq)
    # cuts: a few random ones, some of them one frame apart, none further apart than the bank's batch
    cuts = {0, frames}
    for c in rng.integers(1, frames, size=int(rng.integers(2, 9))):
        cuts.add(int(c))
        if rng.integers(0, 3) == 0:
            cuts.add(min(frames, int(c) + 1))
    cuts = sorted(cuts)
    fine = [cuts[0]]
    for c in cuts[1:]:
        while c - fine[-1] > 256:
            fine.append(fine[-1] + 256)
        fine.append(c)
    recs, debs, edges, peaks = [], {lid: [] for lid in lids}, {lid: [] for lid in lids}, []
    text = {lid: "" for lid in lids}
    for a, b in zip(fine[:-1], fine[1:]):
        assert bank.process_host(iq[a:b]) == b - a
        recs.append(bank.read_frame_records(0))
        for lid in lids:
            debs[lid].append(bank.read_keying_bits(0, lid))
            e = bank.read_edges(0, lid)
            edges[lid].append(np.stack([e["frame"].astype(np.int64), e["state"].astype(np.int64)], axis=1))  # (bank frame numbers: the stream's)
            text[lid] += bank.read_text(0, lid)
        for c in range(bank.last_batch_chunks):
            p, cnt, fr = bank.read_peaks(0, c)
            assert cnt == len(p)
            peaks.append((a + fr, p))
    got = np.concatenate(recs)
    for f in ["min_mean", "variance", "dev_in", "nf_in", "noise_dev", "noise_floor", "peak_thr", "listen_thr"]:
        assert _bits_equal(got[f], out["frames"][f]), (seed, f)
    for lid in lids:
        deb = out["deb"][:, lid]
        assert np.array_equal(np.concatenate(debs[lid]), deb), (seed, lid)
        d8 = deb.astype(np.int8)
        trans = np.flatnonzero(np.diff(np.concatenate([[0], d8])) != 0)
        e = np.concatenate(edges[lid]) if edges[lid] else np.zeros((0, 2), np.int64)
        assert np.array_equal(e[:, 0], trans) and np.array_equal(e[:, 1], d8[trans]), (seed, lid)
        assert text[lid] == ref.text(lid), (seed, lid)
        assert np.array_equal(bank.read_decoder_state(0, lid), ref.decoder_state(lid)), (seed, lid)
    assert [f for f, _ in peaks] == list(out["peak_frames"])
    assert [p for _, p in peaks] == out["peaks"]
    bank.close()
