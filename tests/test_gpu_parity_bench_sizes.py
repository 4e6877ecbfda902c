"""GPU parity at the sizes bench.py runs ("do this" 1): the HIP path through its C ABI against the CPU
oracle, bit for bit, on the benchmark's own generator (synth.make_band_torch) and batch geometry.

  config 3   N = 16384, 256 listeners, 2048-frame batches (two of them: everything carried between batches) and the
             bench's default 8192-frame batch
  config 2   N = 4096, 16 listeners, 4096-frame batches
  config 5   8 channels x N = 8192 x 16 listeners, 2048-frame batches (the per-GPU share), eager and as hipGraph replays

What is compared per batch: frame records (min mean, variance, dB inputs, rolling means, thresholds), every listener's
debounced keying bits, keying edges, decoded text and decoder state (12 float64), every completed cumulation and its
peak list (bins, values, float64 -> int frequencies).  The oracle needs 30 MS/s per core; bands run on threads of their
own (ctypes releases the GIL).
"""
from concurrent.futures import ThreadPoolExecutor

import os

import numpy as np
import pytest

from oracle import oracle as orc
from sdrainer_amd import synth

pytestmark = pytest.mark.gpu

REC_FIELDS = ["min_mean", "variance", "dev_in", "nf_in", "noise_dev", "noise_floor", "peak_thr", "listen_thr"]


@pytest.fixture(scope="module")
def capi():
    from sdrainer_amd import capi as c
    c.load()
    return c


def _bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape
    u = {4: np.uint32, 8: np.uint64, 1: np.uint8}[a.dtype.itemsize]
    return np.array_equal(a.view(u), b.view(u))


def _transitions(deb_col, a, e):
    deb = deb_col.astype(np.int8)
    trans = np.flatnonzero(np.diff(np.concatenate([[0], deb])) != 0)
    trans = trans[(trans >= a) & (trans < e)]
    return trans, deb[trans]


def _peaks_of(res, ch):
    return [tuple(int(p[k]) if k != "signal_value" else float(p[k]) for k in
                  ("from", "to", "from_frequency", "to_frequency", "signal_frequency", "signal_value", "signal_bin"))
            for p in res["peaks"][ch["first_peak"]:ch["first_peak"] + ch["n_peaks"]]]


def _run_oracle(rate, n, edge, bins_per_band, iq_per_band, centers):
    """One oracle receiver per band over the whole stream, bands in parallel."""
    refs = []
    for bins, cf in zip(bins_per_band, centers):
        r = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=cf)
        for b in bins:
            r.attach(int(b))
        refs.append(r)
    with ThreadPoolExecutor(max(1, min(len(refs), 16))) as ex:
        outs = list(ex.map(lambda ri: ri[0].process(ri[1]), zip(refs, iq_per_band)))
    return refs, outs


def _check_batch_polled(res, outs, a, e, tones, text, n_bands):
    """One polled batch (sdr_poll: what bench.py's consumer thread receives) against the oracle's whole-run output."""
    assert res["first_frame"] == a and res["n_frames"] == e - a
    assert res["runes_dropped"] == 0 and res["edges_dropped"] == 0
    by = {(int(r["band"]), int(r["listener"])): r for r in res["listeners"]}
    n_edges = 0
    for band in range(n_bands):
        out = outs[band]
        for lid in range(tones):
            trans, states = _transitions(out["deb"][:, lid], a, e)
            r = by.get((band, lid))
            if r is None:
                assert len(trans) == 0, f"band {band} listener {lid}: edges missing"
                continue
            ed = res["edges"][r["first_edge"]:r["first_edge"] + r["n_edges"]]
            assert np.array_equal(ed["frame"], trans) and np.array_equal(ed["state"], states), f"band {band} listener {lid} edges"
            n_edges += len(trans)
            text[band][lid] += "".join(chr(int(x)) for x in res["runes"][r["first_rune"]:r["first_rune"] + r["n_runes"]])
    n_peaks = 0
    seen = set()
    for ch in res["chunks"]:
        band = int(ch["band"])
        out = outs[band]
        gc = list(out["peak_frames"]).index(int(ch["frame"]))
        got = _peaks_of(res, ch)
        assert got == out["peaks"][gc] and ch["peaks_found"] == len(got), f"band {band} peaks of cumulation {gc}"
        n_peaks += len(got)
        seen.add((band, gc))
    want = {(band, gc) for band in range(n_bands) for gc, f in enumerate(outs[band]["peak_frames"]) if a <= f < e}
    assert seen == want
    return n_edges, n_peaks


def _eager_case(capi, rate, n, tones, n_bands, frames, n_batches, free_last, seed, tail_frames=0):
    """n_batches batches of `frames` frames (and, if tail_frames, a shorter one behind them)."""
    import torch

    edge = synth.default_edge_width(n)
    total = frames * n_batches + tail_frames
    dev_iq, bins_per_band, host_iq = [], [], []
    for b in range(n_bands):
        iq, bins, _ = synth.make_band_torch(total, rate, n, tones, seed=seed + 17 * b, device="cuda", free_last_window=free_last)
        dev_iq.append(iq)
        bins_per_band.append(bins)
        host_iq.append(iq.cpu().numpy())
    centers = [14000000 + 100000 * b for b in range(n_bands)]
    refs, outs = _run_oracle(rate, n, edge, bins_per_band, host_iq, centers)
    bank = capi.Bank(rate, n, n_bands=n_bands, edge_width=edge, max_batch_frames=frames, max_listeners=tones, max_peaks=1024)
    bank.set_stream(torch.cuda.current_stream().cuda_stream)
    for b in range(n_bands):
        bank.set_center_frequency(b, centers[b])
        for i, bn in enumerate(bins_per_band[b]):
            assert bank.attach(b, int(bn)) == i
    bank.enable_results(True)
    text = [["" for _ in range(tones)] for _ in range(n_bands)]
    edges = peaks = 0
    spans = [(k * frames, (k + 1) * frames) for k in range(n_batches)] + ([(frames * n_batches, total)] if tail_frames else [])
    for k, (a, e) in enumerate(spans):
        batch = torch.stack([iq[a:e] for iq in dev_iq]).contiguous()  # [band][frame][2N]
        bank.process_device(batch.data_ptr(), e - a)
        res = bank.poll(wait=True)
        assert res["batch_index"] == k
        ne, npk = _check_batch_polled(res, outs, a, e, tones, text, n_bands)
        edges += ne
        peaks += npk
        # what stays on the device: frame records, keying bits, cumulations
        for b in range(n_bands):
            recs = bank.read_frame_records(b)
            for f in REC_FIELDS:
                assert _bits_equal(recs[f], outs[b]["frames"][f][a:e].copy()), f"band {b} batch {k} field {f}"
            for lid in range(tones):
                assert np.array_equal(bank.read_keying_bits(b, lid), outs[b]["deb"][a:e, lid]), f"band {b} listener {lid} batch {k}"
            for c in range(bank.last_batch_chunks):
                _, _, fr = bank.read_peaks(b, c)
                gc = list(outs[b]["peak_frames"]).index(a + fr)
                exact = outs[b]["cumulation"][gc]
                assert _bits_equal(bank.read_cumulation(b, c), exact), f"band {b} cumulation {gc}"
                # the row as the pipeline keeps it (k_peaks.hip: exact where FindPeaks reads it, an upper bound elsewhere):
                # never below the exact cumulation in any bin, equal to it in every bin of every peak and beside its maximum
                os.environ["SDR_READ_CUM_RAW"] = "1"
                try:
                    raw = bank.read_cumulation(b, c)
                finally:
                    del os.environ["SDR_READ_CUM_RAW"]
                assert np.all(raw >= exact), f"band {b} cumulation {gc}: the kept row is below the exact one somewhere"
                pk, _, _ = bank.read_peaks(b, c)
                for p in pk:
                    lo, hi = max(p[0], p[6] - 1), min(p[1], p[6] + 1)
                    assert _bits_equal(raw[p[0]:p[1] + 1], exact[p[0]:p[1] + 1]) and _bits_equal(raw[lo:hi + 1], exact[lo:hi + 1])
    for b in range(n_bands):
        for lid in range(tones):
            assert text[b][lid] == refs[b].text(lid), f"band {b} listener {lid} text"
            assert np.array_equal(bank.read_decoder_state(b, lid), refs[b].decoder_state(lid)), f"band {b} listener {lid} state"
    assert bank.read_drop_counters() == (0, 0)
    assert edges > 20 * tones * n_bands and peaks > 0 and any(len(t) > 0 for row in text for t in row)
    bank.close()


def test_config3_at_bench_size(capi):
    """bench.py's default workload: N = 16384, 256 listeners, two 2048-frame batches (decoder 4 signals per wave,
    80-workgroup window sums, 21-cumulation batches: the geometry the throughput number is quoted on)."""
    _eager_case(capi, 2_000_000, 16384, 256, 1, 2048, 2, True, seed=3000)


def test_config3_at_bench_default_batch(capi):
    """bench.py's default batch for config 3: 8192 frames (82 cumulations per batch, 128-word keying rows, four times
    the edges and runes per delivery), then a short batch behind it for everything that is carried over."""
    _eager_case(capi, 2_000_000, 16384, 256, 1, 8192, 1, True, seed=3100, tail_frames=300)


def test_config2_at_bench_size(capi):
    """bench.py --workload c2: N = 4096, 16 listeners (a decoder wave each), 4096-frame batches."""
    _eager_case(capi, 192_000, 4096, 16, 1, 4096, 2, False, seed=2000)


def test_config5_share_at_bench_size(capi):
    """bench.py --workload c5: 8 channels x 8192 points x 16 listeners per GPU, 2048-frame batches."""
    _eager_case(capi, 2_000_000, 8192, 16, 8, 2048, 2, False, seed=5000)


def test_graph_mode_at_config5_geometry(capi):
    """hipGraph replays at config 5's geometry (8 bands x 8192 x 16 listeners; device-side cursors for 8 bands): two
    replays of sdr_graph_batches() batches of 460 frames - cumulation phase, carry buffer and frame numbering differ
    from batch to batch - against the oracle: delivery (edges, runes, peaks), records, bits, decoder state."""
    import torch

    rate, n, tones, n_bands, per = 2_000_000, 8192, 16, 8, 460
    edge = synth.default_edge_width(n)
    bank = capi.Bank(rate, n, n_bands=n_bands, edge_width=edge, max_batch_frames=per, max_listeners=tones, max_peaks=256)
    K = bank.graph_batches
    total = 2 * K * per
    dev_iq, bins_per_band, host_iq = [], [], []
    for b in range(n_bands):
        iq, bins, _ = synth.make_band_torch(total, rate, n, tones, seed=5500 + 17 * b, device="cuda")
        dev_iq.append(iq)
        bins_per_band.append(bins)
        host_iq.append(iq.cpu().numpy())
    centers = [7000000 + 50000 * b for b in range(n_bands)]
    refs, outs = _run_oracle(rate, n, edge, bins_per_band, host_iq, centers)
    stream = torch.cuda.Stream()
    bank.set_stream(stream.cuda_stream)
    for b in range(n_bands):
        bank.set_center_frequency(b, centers[b])
        for i, bn in enumerate(bins_per_band[b]):
            assert bank.attach(b, int(bn)) == i
    bank.enable_results(True)
    bank.graph_capture(per)
    batches = [torch.stack([iq[k * per:(k + 1) * per] for iq in dev_iq]).contiguous() for k in range(2 * K)]
    torch.cuda.synchronize()
    text = [["" for _ in range(tones)] for _ in range(n_bands)]
    delivered = 0
    for rep in range(2):
        bank.graph_launch([batches[rep * K + k].data_ptr() for k in range(K)])
        for k in range(K):
            res = bank.poll(wait=True)
            a = (rep * K + k) * per
            assert res["batch_index"] == delivered
            _check_batch_polled(res, outs, a, a + per, tones, text, n_bands)
            delivered += 1
    bank.sync()
    assert bank.total_frames == total
    for b in range(n_bands):
        recs = bank.read_frame_records(b)
        for f in REC_FIELDS:
            assert _bits_equal(recs[f], outs[b]["frames"][f][total - per:].copy()), f"band {b} field {f}"
        for lid in range(tones):
            assert text[b][lid] == refs[b].text(lid), f"band {b} listener {lid}"
            assert np.array_equal(bank.read_keying_bits(b, lid), outs[b]["deb"][total - per:, lid])
            assert np.array_equal(bank.read_decoder_state(b, lid), refs[b].decoder_state(lid))
    # attaching or detaching invalidates the capture (include/sdrainer_hip.h)
    bank.detach(0, 3)
    with pytest.raises(capi.SdrError) as ei:
        bank.graph_launch([batches[0].data_ptr()] * K)
    assert ei.value.code == capi.ERR_STATE
    bank.close()


def test_listeners_bound_inside_a_batch_at_config3_size(capi):
    """The deferred listen half at config 3's geometry (N = 16384, a 2048-frame batch): twenty listeners bound one per
    cumulation boundary INSIDE the batch (sdr_attach_at: gather from the retained psd rows, decoders starting mid-batch)
    beside 64 that were there from the start, against the oracle run cumulation by cumulation with a plain attach at
    every boundary (rx/receiver.go:404-426)."""
    import torch

    rate, n, frames = 2_000_000, 16384, 2048
    early, late = 64, 20
    edge = synth.default_edge_width(n)
    iq, bins, _ = synth.make_band_torch(frames, rate, n, early + late, seed=3300, device="cuda", free_last_window=True)
    host = iq.cpu().numpy()
    ref = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=14000000)
    for b in bins[:early]:
        ref.attach(int(b))
    starts = {}
    outs, pos = [], 0
    for j in range(late):
        boundary = 100 * (j + 1)
        outs.append((pos, ref.process(host[pos:boundary])))
        pos = boundary
        starts[ref.attach(int(bins[early + j]))] = boundary
    outs.append((pos, ref.process(host[pos:])))

    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=frames, max_listeners=early + late, max_peaks=1024)
    bank.set_stream(torch.cuda.current_stream().cuda_stream)
    bank.set_center_frequency(0, 14000000)
    for i, b in enumerate(bins[:early]):
        assert bank.attach(0, int(b)) == i
    bank.enable_results(True)
    bank.defer_listen(True)
    bank.process_device(iq.data_ptr(), frames)
    pk = bank.poll_peaks(wait=True)
    assert [int(c["frame"]) for c in pk["chunks"]] == list(range(99, frames, 100))
    for lid, s in sorted(starts.items()):
        assert bank.attach_at(0, int(bins[lid]), s) == lid
    bank.process_listen()
    res = bank.poll(wait=True)
    assert res["n_frames"] == frames and res["runes_dropped"] == 0 and res["edges_dropped"] == 0
    by = {int(r["listener"]): r for r in res["listeners"]}
    n_edges = 0
    for lid in range(early + late):
        s = starts.get(lid, 0)
        want, last = [], 0
        for base, out in outs:
            if out["deb"].shape[1] <= lid or base + out["deb"].shape[0] <= s:
                continue
            deb = out["deb"][:, lid].astype(np.int8)
            idx = np.flatnonzero(np.diff(np.concatenate([[last], deb])) != 0)
            want += [(base + int(i), int(deb[i])) for i in idx if base + i >= s]
            last = int(deb[-1])
        r = by.get(lid)
        got = [] if r is None else [(int(x["frame"]), int(x["state"])) for x in res["edges"][r["first_edge"]:r["first_edge"] + r["n_edges"]]]
        assert got == want, f"listener {lid} (from frame {s})"
        text = "" if r is None else "".join(chr(int(x)) for x in res["runes"][r["first_rune"]:r["first_rune"] + r["n_runes"]])
        assert text == ref.text(lid), f"listener {lid} text"
        assert np.array_equal(bank.read_decoder_state(0, lid), ref.decoder_state(lid)), f"listener {lid} state"
        n_edges += len(got)
    assert n_edges > 20 * early
    bank.close()
