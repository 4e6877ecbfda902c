"""GPU parity: libsdrainer_hip.so (through its C ABI) against the CPU oracle on the same seeded IQ.

Bars (BASELINE.json north_star): peak bins and keying edges bit-exact; FFT magnitudes within 1e-5
relative.  The HIP path computes the FFT in float64 with the reference's own butterfly order, so the
tests below demand MORE than the north star: spectrum / psd / every threshold BIT-IDENTICAL to the
oracle (np.array_equal on the float32 / float64 bit patterns), and the 1e-5 magnitude bound is
checked on top against numpy's float64 FFT.
"""
import json
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
    assert a.dtype == b.dtype and a.shape == b.shape
    u = {4: np.uint32, 8: np.uint64, 1: np.uint8}[a.dtype.itemsize]
    return np.array_equal(a.view(u), b.view(u))


REC_FIELDS = ["min_mean", "variance", "dev_in", "nf_in", "noise_dev", "noise_floor", "peak_thr", "listen_thr"]


def _assert_records_equal(got, ref, what=""):
    for f in REC_FIELDS:
        assert _bits_equal(got[f], ref[f]), f"{what} frame record field {f} differs"


def _peak_tuple_equal(g, r):
    return g == r


@pytest.mark.parametrize("n,rate,tones", [(512, 48000, 4), (1024, 96000, 6), (2048, 192000, 8), (4096, 192000, 16),
                                          (8192, 2000000, 16), (16384, 2000000, 32)])
def test_spectrum_psd_bit_exact(capi, n, rate, tones):
    frames = 6
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=100 + n)
    bank = capi.Bank(rate, n, max_batch_frames=8, max_listeners=4)
    assert bank.process_host(iq) == frames
    x = iq[:, 0::2].astype(np.float64) + 1j * iq[:, 1::2].astype(np.float64)
    for f in range(frames):
        sp, psd = bank.read_spectrum(0, f)
        sp_ref, psd_ref = orc.iq_to_spectrum_and_psd(iq[f])
        assert _bits_equal(psd, psd_ref), f"psd frame {f}"
        assert _bits_equal(sp, sp_ref), f"spectrum frame {f}"
        # north-star bound: magnitudes within 1e-5 relative of the mathematical DFT
        mag_ref = np.abs(np.fft.fftshift(np.fft.fft(x[f])))
        mag = np.sqrt(psd.astype(np.float64))
        assert np.max(np.abs(mag - mag_ref) / np.maximum(mag_ref, 1e-30)) < 1e-5
    bank.close()


@pytest.mark.parametrize("n,rate,tones,frames", [(512, 48000, 4, 330), (4096, 192000, 16, 250)])
def test_receiver_run_bit_exact(capi, n, rate, tones, frames):
    """Whole frame loop: thresholds, listener traces, text, decoder state, cumulation, peaks."""
    iq, bins, key = synth.make_band(frames, rate, n, tones, seed=7 + n)
    edge = synth.default_edge_width(n)
    ref = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=7020000)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=512, max_listeners=tones + 2, trace=True,
                     max_peaks=256)
    bank.set_center_frequency(0, 7020000)
    lids = [bank.attach(0, int(b)) for b in bins]
    rids = [ref.attach(int(b)) for b in bins]
    assert lids == rids
    out = ref.process(iq)
    assert bank.process_host(iq) == frames
    _assert_records_equal(bank.read_frame_records(0), out["frames"])
    for lid in lids:
        v, r, d = bank.read_trace(0, lid)
        assert _bits_equal(v, out["values"][:, lid].copy())
        assert np.array_equal(r, out["raw"][:, lid]) and np.array_equal(d, out["deb"][:, lid])
        assert np.array_equal(bank.read_keying_bits(0, lid), out["deb"][:, lid])
        # edges = transitions of the debounced stream (the input of Decoder.Tick)
        deb = out["deb"][:, lid].astype(np.int8)
        trans = np.flatnonzero(np.diff(np.concatenate([[0], deb])) != 0)
        e = bank.read_edges(0, lid)
        assert np.array_equal(e["frame"], trans) and np.array_equal(e["state"], deb[trans])
        assert np.array_equal(bank.read_decoder_state(0, lid), ref.decoder_state(lid))
        assert bank.read_text(0, lid) == ref.text(lid)
    assert bank.last_batch_chunks == out["n_chunks"] == frames // 100
    for c in range(out["n_chunks"]):
        assert _bits_equal(bank.read_cumulation(0, c), out["cumulation"][c])
        peaks, count, fr = bank.read_peaks(0, c)
        assert fr == out["peak_frames"][c] and count == len(out["peaks"][c])
        assert peaks == out["peaks"][c]
        assert len(peaks) >= tones  # every keyed tone is found
    bank.close()


def test_batch_split_invariance_and_carry(capi):
    """Any split of the stream into process calls gives the same results (rolling means, cumulation
    carry, debouncer / decoder state all live in HBM between calls)."""
    n, rate, tones, frames = 512, 48000, 3, 437
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=99)
    ref = orc.Receiver(rate, n, 70)
    for b in bins:
        ref.attach(int(b))
    out = ref.process(iq)
    bank = capi.Bank(rate, n, max_batch_frames=256, max_listeners=8, trace=True, max_peaks=64)
    lids = [bank.attach(0, int(b)) for b in bins]
    cuts = [0, 1, 60, 61, 99, 100, 101, 256, 300, 301, 437]
    recs, debs, peaks, text = [], {lid: [] for lid in lids}, [], {lid: "" for lid in lids}
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert bank.process_host(iq[a:b]) == b - a
        recs.append(bank.read_frame_records(0))
        for lid in lids:
            debs[lid].append(bank.read_keying_bits(0, lid))
            text[lid] += bank.read_text(0, lid)
        for c in range(bank.last_batch_chunks):
            p, cnt, fr = bank.read_peaks(0, c)
            peaks.append((a + fr, p))
    _assert_records_equal(np.concatenate(recs), out["frames"])
    for lid in lids:
        assert np.array_equal(np.concatenate(debs[lid]), out["deb"][:, lid])
        assert text[lid] == ref.text(lid)
    assert [f for f, _ in peaks] == list(out["peak_frames"])
    assert [p for _, p in peaks] == out["peaks"]
    bank.close()


def test_multi_band_independent(capi):
    n, rate, frames, B = 1024, 96000, 130, 3
    bands = [synth.make_band(frames, rate, n, 4, seed=500 + b) for b in range(B)]
    bank = capi.Bank(rate, n, n_bands=B, max_batch_frames=256, max_listeners=8, trace=True, max_peaks=64)
    refs = []
    for b, (iq, bins, _) in enumerate(bands):
        r = orc.Receiver(rate, n, synth.default_edge_width(n))
        for bn in bins[: b + 1]:  # different listener counts per band
            bank.attach(b, int(bn))
            r.attach(int(bn))
        refs.append(r)
    allq = np.stack([iq for iq, _, _ in bands])
    assert bank.process_host(allq) == frames
    for b in range(B):
        out = refs[b].process(bands[b][0])
        _assert_records_equal(bank.read_frame_records(b), out["frames"], f"band {b}")
        for lid in range(b + 1):
            assert np.array_equal(bank.read_keying_bits(b, lid), out["deb"][:, lid])
            assert bank.read_text(b, lid) == refs[b].text(lid)
        p, cnt, fr = bank.read_peaks(b, 0)
        assert p == out["peaks"][0] and fr == 99
    bank.close()


def test_recorded_streams_through_device_decoder(capi, golden_dir):
    """The reference's nine recorded on/off streams (cw/decode_test.go:177-213) keyed onto a tone and
    run through the whole GPU path must decode to the reference's expected strings."""
    d = os.path.join(golden_dir, "cw_streams")
    spec = json.load(open(os.path.join(d, "expected.json")))
    n, rate = spec["block_size"], spec["sample_rate"]
    bank = capi.Bank(rate, n, max_batch_frames=4096, max_listeners=2, trace=True, find_peaks=False)
    rng = np.random.default_rng(2024)
    b = 300
    fft_bin = (b + n // 2) % n
    tone = 0.1 * np.exp(2j * np.pi * fft_bin * np.arange(n) / n)

    def frames_for(bits):
        x = bits[:, None] * tone[None, :] + 1e-3 * (rng.standard_normal((len(bits), n)) +
                                                    1j * rng.standard_normal((len(bits), n)))
        iq = np.empty((len(bits), 2 * n), np.float32)
        iq[:, 0::2], iq[:, 1::2] = x.real, x.imag
        return iq

    bank.process_host(frames_for(np.zeros(80)))  # let both 60-frame rolling means fill up
    for fname, expected in spec["cases"]:
        bits = np.array([1.0 if ln.strip() == "1" else 0.0 for ln in open(os.path.join(d, fname)).read().split("\n")
                         if ln.strip() != ""])
        lid = bank.attach(0, b)  # a fresh listener = new decoder + Reset (rx/listener.go:84-94)
        bank.process_host(frames_for(bits))
        _, raw, deb = bank.read_trace(0, lid)
        assert np.array_equal(deb, bits.astype(np.uint8)), fname
        bank.listener_stop(0, lid)  # decoder.stop()
        assert bank.read_text(0, lid) == expected, fname
        bank.detach(0, lid)
    bank.close()


def test_push_iq_error_behaviour(capi):
    """rx/receiver.go:315-334: wrong rate / wrong size / full queue are reported, nothing is processed."""
    n, rate = 512, 48000
    bank = capi.Bank(rate, n, max_batch_frames=4)
    frame = np.zeros(2 * n, np.float32)
    assert bank.push_iq(0, 44100, frame) == capi.ERR_BAD_RATE
    assert bank.push_iq(0, rate, frame[:-2]) == capi.ERR_BAD_SIZE
    assert bank.staged_frames(0) == 0
    for _ in range(4):
        assert bank.push_iq(0, rate, frame + 1e-3) == capi.OK
    assert bank.push_iq(0, rate, frame) == capi.ERR_WOULD_DROP
    assert bank.staged_frames(0) == 4
    with pytest.raises(capi.SdrError):
        capi.Bank(rate, 500)  # not a power of two
    with pytest.raises(capi.SdrError):
        capi.Bank(rate, 512, edge_width=252)  # windowSize would be 0 -> NaN in the reference
    bank.close()


def test_listener_pool_and_controls(capi):
    n, rate = 512, 48000
    iq, bins, _ = synth.make_band(200, rate, n, 2, seed=11)
    bank = capi.Bank(rate, n, max_batch_frames=256, max_listeners=2, trace=True)
    a = bank.attach(0, int(bins[0]))
    b = bank.attach(0, int(bins[1]))
    with pytest.raises(capi.SdrError) as ei:
        bank.attach(0, 100)
    assert ei.value.code == capi.ERR_NO_SLOT
    bank.detach(0, a)
    assert bank.listener_count(0) == 1
    c = bank.attach(0, int(bins[0]))
    assert c == a  # released slot is reused
    # setters apply at the next batch: threshold, edge width, debounce
    ref = orc.Receiver(rate, n, 50, 9.0, 1)
    ref.attach(int(bins[0]))
    ref.attach(int(bins[1]))
    bank.set_edge_width(50)
    bank.set_peak_threshold(0, 9.0)
    out = ref.process(iq)
    bank.process_host(iq)
    _assert_records_equal(bank.read_frame_records(0), out["frames"])
    p, _, _ = bank.read_peaks(0, 1)
    assert p == out["peaks"][1]
    bank.close()


def test_debounce_threshold_3(capi):
    n, rate = 512, 48000
    iq, bins, _ = synth.make_band(260, rate, n, 2, seed=21)
    ref = orc.Receiver(rate, n, 70, 15.0, 3)
    bank = capi.Bank(rate, n, signal_debounce=3, max_batch_frames=512, max_listeners=4, trace=True)
    for b in bins:
        ref.attach(int(b))
        bank.attach(0, int(b))
    out = ref.process(iq)
    bank.process_host(iq)
    for lid in range(2):
        _, r, d = bank.read_trace(0, lid)
        assert np.array_equal(r, out["raw"][:, lid]) and np.array_equal(d, out["deb"][:, lid])
        assert bank.read_text(0, lid) == ref.text(lid)
    bank.close()


@pytest.mark.parametrize("debounce", [1, 2, 5, 9, 70])
def test_debounce_thresholds_weak_signals_and_carry(capi, debounce):
    """The debouncer per 64-tick word and the staged decoder (cw_stages.h) against the oracle's literal Debounce + Tick:
    signals weak enough for the raw states to glitch, thresholds from the pass-through (1) to longer than a word (70),
    four uneven batches, one of a single frame (the debouncer's count, the decoder's clocks and the current character are
    carried)."""
    n, rate, frames = 512, 48000, 700
    iq, bins, _ = synth.make_band(frames, rate, n, 4, seed=77 + debounce, amplitude=synth.TONE_AMPLITUDE * 0.0027)
    ref = orc.Receiver(rate, n, 70, 15.0, debounce)
    bank = capi.Bank(rate, n, signal_debounce=debounce, max_batch_frames=512, max_listeners=4, trace=True)
    for b in bins:
        ref.attach(int(b))
        bank.attach(0, int(b))
    glitches = 0
    for lo, hi in ((0, 130), (130, 131), (131, 517), (517, 700)):
        out = ref.process(iq[lo:hi])
        bank.process_host(iq[lo:hi])
        for lid in range(4):
            _, r, d = bank.read_trace(0, lid)
            assert np.array_equal(r, out["raw"][:, lid]) and np.array_equal(d, out["deb"][:, lid]), (debounce, lo, lid)
            glitches += int(np.count_nonzero(np.diff(out["raw"][:, lid].astype(np.int8))))
    for lid in range(4):
        assert bank.read_text(0, lid) == ref.text(lid)
    assert glitches > 450  # (the raw states chatter - clean keying of these frames has about 240 transitions: that is what the test is for)
    bank.close()


def test_device_resident_input_and_profile(capi):
    """sdr_process_device: IQ already in HBM (torch tensor), run on torch's current stream."""
    import torch

    n, rate, frames = 4096, 192000, 64
    iq, bins, _ = synth.make_band(frames, rate, n, 8, seed=31)
    bank = capi.Bank(rate, n, max_batch_frames=64, max_listeners=8)
    bank.set_stream(torch.cuda.current_stream().cuda_stream)
    t = torch.from_numpy(iq).cuda()
    bank.profile_enable(True)
    bank.process_device(t.data_ptr(), frames)
    bank.sync()
    prof = bank.profile_read()
    assert prof["k_fft_psd"][1] == 1 and prof["k_fft_psd"][0] > 0
    ref = orc.Receiver(rate, n, synth.default_edge_width(n))
    out = ref.process(iq)
    _assert_records_equal(bank.read_frame_records(0), out["frames"])
    # the device pointer must be 16-byte aligned (frames are staged into LDS 16 bytes per lane)
    with pytest.raises(capi.SdrError) as e:
        bank.process_device(t.data_ptr() + 8, 1)
    assert e.value.code == capi.ERR_BAD_ARG
    bank.close()


def test_audio_path_bit_exact(capi):
    """cw/audio.go chain (BASELINE config 1): Goertzel magnitudes, states and text vs the oracle."""
    sr = 48000
    ref = orc.AudioDemodulator(700.0, sr)
    bs = ref.blocksize
    keying = orc.generate_stream(sr, bs, 20, "cq de dl1abc")
    env = np.repeat(keying, bs).astype(np.float32)
    t = np.arange(env.size) / sr
    rng = np.random.default_rng(5)
    sig = (0.8 * np.cos(2 * np.pi * 700.0 * t) * env + 0.01 * rng.standard_normal(env.size)).astype(np.float32)
    for scale in (0.0, 1.0):
        ref = orc.AudioDemodulator(700.0, sr)
        ref.set_scale(scale)
        ab = capi.AudioBank(2, 700.0, sr, max_blocks=4096)
        assert ab.blocksize == bs == 207
        ab.set_scale(scale)
        # ragged writes: remainders carry over to the next call (cw/audio.go:175-179)
        cuts = [0, 1000, 1001, 50000, env.size]
        mags, raws, debs = [], [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            ab.write(np.stack([sig[a:b], sig[a:b][::-1] * 0]))
            m, r, d = ab.read_trace(0)
            mags.append(m), raws.append(r), debs.append(d)
        rm, rr, rd = ref.write(sig)
        ref.close()
        ab.close()
        assert _bits_equal(np.concatenate(mags), rm)
        assert np.array_equal(np.concatenate(raws), rr) and np.array_equal(np.concatenate(debs), rd)
        assert ab.read_text(0) == ref.text() == "cq de dl1abc"
        assert ab.read_text(1) == ""
        ab.close_handle()


def test_full_size_properties(capi):
    """BASELINE config 3 geometry through size-independent properties - keying round trip (what was keyed is
    what is detected), every tone found by the peak scan at its bin, determinism (two runs give identical bits).
    (The bit-for-bit comparison with the oracle at this geometry and at the benchmarked batch sizes is
    tests/test_gpu_parity_bench_sizes.py; the oracle needs about a second per 2048 frames.)"""
    import torch

    n, rate, tones, frames = 16384, 2000000, 256, 1024
    iq, bins, key = synth.make_band_torch(frames, rate, n, tones, seed=3003, device="cuda", free_last_window=True)
    results = []
    for _ in range(2):
        bank = capi.Bank(rate, n, max_batch_frames=frames, max_listeners=tones, max_peaks=1024)
        bank.set_stream(torch.cuda.current_stream().cuda_stream)
        for b in bins:
            bank.attach(0, int(b))
        bank.process_device(iq.data_ptr(), frames)
        bank.sync()
        bits = np.stack([bank.read_keying_bits(0, l) for l in range(tones)], axis=1)
        peaks = [bank.read_peaks(0, c)[0] for c in range(bank.last_batch_chunks)]
        recs = bank.read_frame_records(0)
        results.append((bits, peaks, recs))
        bank.close()
    bits, peaks, recs = results[0]
    assert np.array_equal(bits, results[1][0]) and peaks == results[1][1]
    assert _bits_equal(recs["listen_thr"], results[1][2]["listen_thr"])
    # after the 60-frame warm-up of the rolling means the detected keying IS the transmitted keying
    # (a noise-only bin crosses listen_thr with probability ~1e-7 per frame: allow a couple of blips)
    assert np.count_nonzero(bits[64:] != key[64:]) <= 2
    assert len(peaks) == frames // 100
    want = set(int(b) for b in bins)
    for chunk in peaks[1:]:
        found = {p[6] for p in chunk}
        assert found <= want  # no spurious peaks: every reported signal bin is a transmitted carrier
        assert len(found) >= 0.9 * len(want)  # a carrier idling in a word gap for most of the 100 frames is skipped


@pytest.mark.parametrize("n,rate,tones,frames,listeners", [(8192, 2000000, 16, 130, 16), (16384, 2000000, 64, 130, 64)])
def test_receiver_run_bit_exact_large_blocks(capi, n, rate, tones, frames, listeners):
    """The big block sizes of BASELINE configs 3-5 through the whole loop (noise chains span 190 tiles)."""
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=n + 5, free_last_window=True)
    edge = synth.default_edge_width(n)
    ref = orc.Receiver(rate, n, edge)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=frames, max_listeners=listeners, max_peaks=512)
    for b in bins[:listeners]:
        ref.attach(int(b))
        bank.attach(0, int(b))
    out = ref.process(iq)
    assert bank.process_host(iq) == frames
    _assert_records_equal(bank.read_frame_records(0), out["frames"])
    for lid in range(listeners):
        assert np.array_equal(bank.read_keying_bits(0, lid), out["deb"][:, lid])
        assert bank.read_text(0, lid) == ref.text(lid)
    assert _bits_equal(bank.read_cumulation(0, 0), out["cumulation"][0])
    assert bank.read_peaks(0, 0)[0] == out["peaks"][0]
    bank.close()


def test_nine_window_geometry(capi):
    """(N - 2*edge) % 10 == 0: the reference's loop never evaluates the tenth window (dsp/fft.go:226-238)."""
    n, rate, edge = 512, 48000, 66  # span 380 = 10 * 38
    iq, bins, _ = synth.make_band(140, rate, n, 3, seed=66, edge_width=edge)
    ref = orc.Receiver(rate, n, edge)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=256, max_listeners=4, max_peaks=64)
    for b in bins:
        ref.attach(int(b))
        bank.attach(0, int(b))
    out = ref.process(iq)
    bank.process_host(iq)
    _assert_records_equal(bank.read_frame_records(0), out["frames"])
    assert bank.read_peaks(0, 0)[0] == out["peaks"][0]
    bank.close()


@pytest.mark.parametrize("n, edge", [(16384, 100), (16384, 101), (8192, 33), (4096, 1), (2048, 3), (1024, 7), (512, 71), (16384, 2243)])
def test_scan_segment_geometries(capi, n, edge):
    """k_psd_scan cuts a row into the reference's windows and pieces of the two edges, and reads each segment by 16-byte
    loads for whole groups of 256 bins plus dword loads for the rest: windows that start on odd bins (the 16-byte loads are
    only dword-aligned then), the widest windows a block size allows (N = 16384 with a narrow edge: more than 1216 bins, the
    kernel's largest instantiation), windows narrower than one group (N <= 2048: dword loads only), edges shorter than a
    piece.  Records, peaks and a whole cumulation row against the oracle; tests/test_forced_paths.py runs the same cases with
    the cumulation's bound forced on, which is what writes the unit counts through the same lane-to-bin map."""
    rate = 48000
    frames = 230 if n <= 4096 else 130
    iq, bins, _ = synth.make_band(frames, rate, n, 3, seed=n + edge, edge_width=edge)
    ref = orc.Receiver(rate, n, edge)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=256, max_listeners=4, max_peaks=64)
    for b in bins[:2]:  # (the third signal has no listener: with the wide tap in use, its column comes from the psd array)
        ref.attach(int(b))
        bank.attach(0, int(b))
    out = ref.process(iq)
    bank.process_host(iq)
    _assert_records_equal(bank.read_frame_records(0), out["frames"])
    assert len(out["peaks"]) >= 1
    for c in range(len(out["peaks"])):
        assert bank.read_peaks(0, c)[0] == out["peaks"][c], c
        assert _bits_equal(bank.read_cumulation(0, c), out["cumulation"][c]), c
    bank.close()


def _nan_equal_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    both_nan = np.isnan(a) & np.isnan(b)  # the sign / payload of a NaN is not part of the contract
    u = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    return np.array_equal(np.where(both_nan, 0, a.view(u)), np.where(both_nan, 0, b.view(u)))


def test_degenerate_inputs_follow_the_reference(capi):
    """Noise-only frames, then an all-zero frame: log10(0) = -Inf enters the running-sum means, which read
    -Inf for 60 frames and NaN forever after the -Inf leaves the window (-Inf - -Inf; SURVEY §7
    'degenerate inputs') — same on both sides, nothing crashes."""
    n, rate = 512, 48000
    rng = np.random.default_rng(8)
    iq = (1e-3 * rng.standard_normal((150, 2 * n))).astype(np.float32)
    iq[70] = 0.0
    ref = orc.Receiver(rate, n, 70)
    bank = capi.Bank(rate, n, max_batch_frames=256, max_listeners=2, trace=True)
    ref.attach(200)
    bank.attach(0, 200)
    out = ref.process(iq, want_spectrum=True)
    bank.process_host(iq)
    got = bank.read_frame_records(0)
    for f in REC_FIELDS:
        assert _nan_equal_bits(got[f], out["frames"][f]), f
    assert np.isfinite(got["listen_thr"][:70]).all()
    assert np.isneginf(got["listen_thr"][70:130]).all() and np.isnan(got["listen_thr"][130:]).all()
    sp, psd = bank.read_spectrum(0, 70)
    assert np.all(psd == 0) and np.all(np.isneginf(sp))
    v, r, d = bank.read_trace(0, 0)
    assert np.array_equal(r, out["raw"][:, 0])
    assert bank.read_text(0, 0) == ref.text(0)
    bank.close()


def test_detach_and_reattach_mid_stream(capi):
    n, rate = 512, 48000
    iq, bins, _ = synth.make_band(300, rate, n, 2, seed=123)
    ref = orc.Receiver(rate, n, 70)
    bank = capi.Bank(rate, n, max_batch_frames=128, max_listeners=2)
    a = bank.attach(0, int(bins[0]))
    ra = ref.attach(int(bins[0]))
    out1 = ref.process(iq[:100])
    bank.process_host(iq[:100])
    assert np.array_equal(bank.read_keying_bits(0, a), out1["deb"][:, ra])
    bank.detach(0, a)
    ref.detach(ra)
    ref.process(iq[100:200])
    bank.process_host(iq[100:200])
    b = bank.attach(0, int(bins[1]))  # slot of the detached listener is reused, with a brand-new decoder
    rb = ref.attach(int(bins[1]))
    assert b == a
    out3 = ref.process(iq[200:])
    bank.process_host(iq[200:])
    assert np.array_equal(bank.read_keying_bits(0, b), out3["deb"][:, rb])
    assert np.array_equal(bank.read_decoder_state(0, b), ref.decoder_state(rb))
    bank.close()


def test_kiwi_snd_payload_unpacked_on_device(capi):
    """Source wire format: raw KiwiSDR SND payloads (17-byte header + big-endian int16 IQ)
    are unpacked in HBM; the result is what the reference's decodeIQBytes + Receiver.run produce."""
    n, rate, frames = 512, 12000, 120
    rng = np.random.default_rng(12)
    iq16 = rng.integers(-3000, 3000, size=(frames, 2 * n)).astype(np.int16)
    t = np.arange(n)
    tone = (12000 * np.exp(2j * np.pi * 40 * t / n))
    iq16[:, 0::2] += tone.real.astype(np.int16)
    iq16[:, 1::2] += tone.imag.astype(np.int16)
    bank = capi.Bank(rate, n, max_batch_frames=128, max_listeners=2, trace=True)
    ref = orc.Receiver(rate, n, 70)
    b = (40 + n // 2) % n
    bank.attach(0, b)
    ref.attach(b)
    # messages of 1, 2, 5 ... frames each, like the websocket delivers them
    ref_iq, f = [], 0
    for k in [1, 2, 5, 12, 40, 60]:
        payload = bytes([0x01] + [7] * 16) + iq16[f:f + k].astype(">i2").tobytes()
        assert bank.push_kiwi_snd(0, rate, payload) == capi.OK
        ref_iq.append(orc.decode_iq_message(payload).reshape(k, 2 * n))
        f += k
    assert f == frames and bank.staged_frames(0) == frames
    # contract errors: partial frame, wrong rate, mixing float frames into a raw batch
    assert bank.push_kiwi_snd(0, rate, bytes(17) + bytes(10)) == capi.ERR_BAD_SIZE
    assert bank.push_kiwi_snd(0, rate + 1, bytes(17) + bytes(4 * n)) == capi.ERR_BAD_RATE
    assert bank.push_iq(0, rate, np.zeros(2 * n, np.float32)) == capi.ERR_STATE
    assert bank.process_staged() == frames
    out = ref.process(np.concatenate(ref_iq), want_spectrum=True)
    for fr in (0, 57, frames - 1):
        sp, psd = bank.read_spectrum(0, fr)
        assert _bits_equal(sp, out["spectrum"][fr]) and _bits_equal(psd, out["psd"][fr])
    _assert_records_equal(bank.read_frame_records(0), out["frames"])
    assert np.array_equal(bank.read_keying_bits(0, 0), out["deb"][:, 0])
    # afterwards the band accepts float frames again
    assert bank.push_iq(0, rate, np.zeros(2 * n, np.float32) + 1e-3) == capi.OK
    bank.close()


def test_config5_geometry(capi):
    """BASELINE config 5's per-GPU share: one bank of 8 channels x 8192-point FFT x 16 listeners each.  Every
    channel is an independent receiver (rx/receiver.go:64-91) and is checked against its own oracle receiver:
    frame records, keying bits, edges, text, cumulation and peaks, over two batches (state carried across)."""
    import torch

    n, rate, tones, B, frames = 8192, 2000000, 16, 8, 136
    edge = synth.default_edge_width(n)
    bands = [synth.make_band(frames, rate, n, tones, seed=5000 + 17 * b) for b in range(B)]
    bank = capi.Bank(rate, n, n_bands=B, edge_width=edge, max_batch_frames=96, max_listeners=tones, max_peaks=256)
    bank.set_stream(torch.cuda.current_stream().cuda_stream)
    refs = []
    for b, (iq, bins, _) in enumerate(bands):
        r = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=14000000 + 100000 * b)
        bank.set_center_frequency(b, 14000000 + 100000 * b)
        for bn in bins:
            assert bank.attach(b, int(bn)) == r.attach(int(bn))
        refs.append(r)
    outs = [refs[b].process(bands[b][0]) for b in range(B)]
    text = [["" for _ in range(tones)] for _ in range(B)]
    n_peaks = 0
    for a, e in ((0, 96), (96, frames)):
        dev = torch.from_numpy(np.stack([iq[a:e] for iq, _, _ in bands])).cuda()  # [band][frame][2N]
        bank.process_device(dev.data_ptr(), e - a)
        bank.sync()
        for b in range(B):
            out = outs[b]
            recs = bank.read_frame_records(b)
            for f in REC_FIELDS:
                assert _bits_equal(recs[f], out["frames"][f][a:e].copy()), f"band {b} frames {a}:{e} field {f}"
            for lid in range(tones):
                assert np.array_equal(bank.read_keying_bits(b, lid), out["deb"][a:e, lid]), f"band {b} listener {lid}"
                deb = out["deb"][:, lid].astype(np.int8)
                trans = np.flatnonzero(np.diff(np.concatenate([[0], deb])) != 0)
                trans = trans[(trans >= a) & (trans < e)]
                ed = bank.read_edges(b, lid)
                assert np.array_equal(ed["frame"], trans) and np.array_equal(ed["state"], deb[trans])
                text[b][lid] += bank.read_text(b, lid)
            for c in range(bank.last_batch_chunks):
                peaks, count, fr = bank.read_peaks(b, c)
                gc = list(out["peak_frames"]).index(a + fr)
                assert _bits_equal(bank.read_cumulation(b, c), out["cumulation"][gc])
                assert peaks == out["peaks"][gc] and count == len(peaks)
                n_peaks += len(peaks)
    for b in range(B):
        for lid in range(tones):
            assert text[b][lid] == refs[b].text(lid), f"band {b} listener {lid}"
            assert np.array_equal(bank.read_decoder_state(b, lid), refs[b].decoder_state(lid))
    assert n_peaks >= B * tones // 2
    bank.close()


def _check_delivery(res, out, a, e, tones, text, ref_frames_base=0):
    """One polled batch covering frames [a, e) against the oracle's whole-run output."""
    assert res["first_frame"] == a and res["n_frames"] == e - a
    by_listener = {int(r["listener"]): r for r in res["listeners"] if r["band"] == 0}
    for lid in range(tones):
        deb = out["deb"][:, lid].astype(np.int8)
        trans = np.flatnonzero(np.diff(np.concatenate([[0], deb])) != 0)
        trans = trans[(trans >= a) & (trans < e)]
        r = by_listener.get(lid)
        if r is None:
            assert len(trans) == 0
            continue
        ed = res["edges"][r["first_edge"]:r["first_edge"] + r["n_edges"]]
        assert np.array_equal(ed["frame"], trans) and np.array_equal(ed["state"], deb[trans])
        text[lid] += "".join(chr(int(x)) for x in res["runes"][r["first_rune"]:r["first_rune"] + r["n_runes"]])
    for ch in res["chunks"]:
        gc = list(out["peak_frames"]).index(int(ch["frame"]))
        got = [tuple(int(p[k]) if k != "signal_value" else float(p[k]) for k in
                     ("from", "to", "from_frequency", "to_frequency", "signal_frequency", "signal_value", "signal_bin"))
               for p in res["peaks"][ch["first_peak"]:ch["first_peak"] + ch["n_peaks"]]]
        assert got == out["peaks"][gc] and ch["peaks_found"] == len(got)


def test_bulk_delivery_poll(capi):
    """sdr_enable_results / sdr_poll: every batch's peaks, edges and runes arrive once, in order, identical to the
    oracle - polled promptly, polled late (more batches in flight than ring sets: the parked path) and with
    buffers that are too small first (nothing is consumed then)."""
    import ctypes as C

    n, rate, tones, frames = 1024, 96000, 6, 1500
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=77)
    edge = synth.default_edge_width(n)
    ref = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=3500000)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=128, max_listeners=tones + 1, max_peaks=128)
    bank.set_center_frequency(0, 3500000)
    for b in bins:
        assert bank.attach(0, int(b)) == ref.attach(int(b))
    out = ref.process(iq)
    bank.enable_results(True)
    assert bank.poll() is None and bank.results_pending == 0
    text = ["" for _ in range(tones)]
    cuts = list(range(0, frames, 125)) + [frames]
    spans = list(zip(cuts[:-1], cuts[1:]))
    delivered = 0
    # phase 1: poll (blocking) after every batch
    for a, e in spans[:3]:
        assert bank.process_host(iq[a:e]) == e - a
        res = bank.poll(wait=True)
        assert res["batch_index"] == delivered
        _check_delivery(res, out, a, e, tones, text)
        delivered += 1
    # phase 2: seven batches without polling (the ring holds six): the oldest is parked on the host, none is lost
    for a, e in spans[3:10]:
        assert bank.process_host(iq[a:e]) == e - a
    assert bank.results_pending == 7
    # a too-small buffer reports what is needed and consumes nothing
    r = capi.Results()
    r.struct_size = C.sizeof(capi.Results)
    rc = bank._L.sdr_poll(bank._h, C.byref(r), 1)
    assert rc == capi.ERR_BAD_SIZE and r.n_edges > 0 and bank.results_pending == 7
    for a, e in spans[3:10]:
        res = bank.poll(wait=True)
        assert res["batch_index"] == delivered
        _check_delivery(res, out, a, e, tones, text)
        delivered += 1
    # phase 3: non-blocking polls while the rest is enqueued
    for a, e in spans[10:]:
        assert bank.process_host(iq[a:e]) == e - a
    pending = [s for s in spans[10:]]
    while pending:
        res = bank.poll(wait=False)
        if res is None:
            res = bank.poll(wait=True)
        a, e = pending.pop(0)
        assert res["batch_index"] == delivered
        _check_delivery(res, out, a, e, tones, text)
        delivered += 1
    assert bank.poll() is None and delivered == len(spans)
    assert res["runes_dropped"] == 0 and res["edges_dropped"] == 0 and bank.read_drop_counters() == (0, 0)
    for lid in range(tones):
        assert text[lid] == ref.text(lid) and len(text[lid]) > 0
        assert bank.read_text(0, lid) == ""  # delivered through sdr_poll only
    bank.close()


def test_drop_counters_are_exposed(capi):
    """Without polling the per-listener text buffer (2048 runes) eventually fills: further runes are counted, not
    lost silently (the reference's io.Writer never drops; the counters tell the host it polled too rarely)."""
    n, rate = 512, 48000
    frames = 4096
    bank = capi.Bank(rate, n, max_batch_frames=frames, max_listeners=1, find_peaks=False)
    b = 200
    fft_bin = (b + n // 2) % n
    tone = 0.1 * np.exp(2j * np.pi * fft_bin * np.arange(n) / n)
    rng = np.random.default_rng(1)
    # fastest legal keying: two frames on, two off = a stream of dits ("e" after "e"); one rune every ~8 frames
    bits = (np.arange(frames) % 4 < 2).astype(np.float64)
    x = bits[:, None] * tone[None, :] + 1e-3 * (rng.standard_normal((frames, n)) + 1j * rng.standard_normal((frames, n)))
    iq = np.empty((frames, 2 * n), np.float32)
    iq[:, 0::2], iq[:, 1::2] = x.real, x.imag
    bank.attach(0, b)
    for _ in range(24):  # about one rune per 32 frames
        bank.process_host(iq)
    runes, edges = bank.read_drop_counters()
    kept = len(bank.read_text(0, 0))
    assert edges == 0 and kept <= 2048
    assert runes > 0 and kept == 2048, (runes, kept)
    bank.close()


def test_scope_tap(capi):
    """scope.Scope tap (scope/scope.go:14-37): the "spectrum" frame per completed cumulation (rx/receiver.go:428-457)
    and the "demod" frames per listener per frame (cw/spectral.go:56-81) against the oracle's traces; inactive
    (NullScope) unless the bank was created with trace."""
    n, rate, tones, frames = 1024, 96000, 5, 230
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=4711)
    edge = synth.default_edge_width(n)
    ref = orc.Receiver(rate, n, edge)
    quiet = capi.Bank(rate, n, edge_width=edge, max_batch_frames=256, max_listeners=tones)
    assert not quiet.scope_active
    quiet.process_host(iq)
    with pytest.raises(capi.SdrError) as ei:
        quiet.scope_spectral_frame(0, 0)
    assert ei.value.code == capi.ERR_STATE
    quiet.close()
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=256, max_listeners=tones, trace=True)
    assert bank.scope_active
    for b in bins[1:]:  # pool order = attach order: the first listener's bin is the frequency marker
        bank.attach(0, int(b))
        ref.attach(int(b))
    out = ref.process(iq)
    assert bank.process_host(iq) == frames
    for c in range(2):
        hdr, vals = bank.scope_spectral_frame(0, c)
        assert hdr["frame"] == 100 * (c + 1) - 1 and hdr["from_frequency"] == 0.0 and hdr["to_frequency"] == 1.0
        assert hdr["signal_bin"] == float(bins[1]) and hdr["n_values"] == n
        assert hdr["threshold"] == float(out["frames"]["peak_thr"][100 * (c + 1) - 1])
        assert np.array_equal(vals, out["cumulation"][c].astype(np.float64) * (1.0 / 100.0))
    for lid in range(tones - 1):
        fr = bank.scope_demod_frames(0, lid)
        assert len(fr) == frames
        assert np.array_equal(fr["threshold"], out["frames"]["listen_thr"].astype(np.float64))
        assert np.array_equal(fr["value"], out["values"][:, lid].astype(np.float64))
        assert np.array_equal(fr["state"], np.where(out["raw"][:, lid] != 0, 100.0, -1.0))
        assert np.array_equal(fr["debounced"], np.where(out["deb"][:, lid] != 0, 80.0, -1.0))
    bank.close()


def test_graph_mode_bit_exact(capi):
    """sdr_graph_*: the steady state captured as one hipGraph (BASELINE config 5's "hipGraph-captured steady
    state").  Two replays of sdr_graph_batches() batches each - batch length 130, so the cumulation phase, the
    carry buffer and the frame numbering differ from batch to batch and from replay to replay, all of it read from
    the device-side cursors - must deliver exactly what the oracle computes for the whole stream."""
    import torch

    n, rate, tones, per = 1024, 96000, 5, 130
    edge = synth.default_edge_width(n)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=per, max_listeners=tones, max_peaks=128)
    K = bank.graph_batches
    frames = 2 * K * per
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=808)
    ref = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=10100000)
    bank.set_center_frequency(0, 10100000)
    stream = torch.cuda.Stream()
    bank.set_stream(stream.cuda_stream)
    for b in bins:
        assert bank.attach(0, int(b)) == ref.attach(int(b))
    out = ref.process(iq)
    bank.enable_results(True)
    with pytest.raises(capi.SdrError):  # nothing captured yet
        bank.graph_launch([0] * K)
    bank.graph_capture(per)
    dev = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    text = ["" for _ in range(tones)]
    delivered = 0
    for rep in range(2):
        ptrs = [dev[(rep * K + k) * per].data_ptr() for k in range(K)]
        bank.graph_launch(ptrs)
        with pytest.raises(capi.SdrError) as ei:  # eager calls are refused while a graph is captured
            bank.process_device(ptrs[0], per)
        assert ei.value.code == capi.ERR_STATE
        for k in range(K):
            res = bank.poll(wait=True)
            a = (rep * K + k) * per
            assert res["batch_index"] == delivered
            _check_delivery(res, out, a, a + per, tones, text)
            delivered += 1
    bank.sync()
    assert bank.total_frames == frames and bank.last_batch_frames == per
    _assert_records_equal(bank.read_frame_records(0), out["frames"][frames - per:])
    for lid in range(tones):
        assert text[lid] == ref.text(lid) and len(text[lid]) > 0
        assert np.array_equal(bank.read_keying_bits(0, lid), out["deb"][frames - per:, lid])
        assert np.array_equal(bank.read_decoder_state(0, lid), ref.decoder_state(lid))
    # back to eager processing after the release
    bank.graph_release()
    bank.process_device(dev[0].data_ptr(), per)
    bank.sync()
    bank.close()


def test_graph_refuses_replay_after_results_or_find_peaks_changed(capi):
    """Round 4's advice: the captured graphs bake in whether results are delivered (the packing kernels are nodes or they
    are not) and whether peaks are searched (refinement / peak-scan nodes).  sdr_enable_results or sdr_set_find_peaks
    after the capture must make sdr_graph_launch refuse (SDR_ERR_STATE: capture again) instead of publishing batches no
    kernel fills; after a new capture the replays deliver again."""
    import torch

    n, rate, tones, per = 1024, 96000, 3, 120
    edge = synth.default_edge_width(n)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=per, max_listeners=tones, max_peaks=128)
    K = bank.graph_batches
    iq, bins, _ = synth.make_band(2 * K * per, rate, n, tones, seed=909)
    ref = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=10100000)
    bank.set_center_frequency(0, 10100000)
    stream = torch.cuda.Stream()
    bank.set_stream(stream.cuda_stream)
    for b in bins:
        assert bank.attach(0, int(b)) == ref.attach(int(b))
    out = ref.process(iq)
    dev = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    bank.graph_capture(per)  # captured WITHOUT results
    bank.enable_results(True)
    ptrs = [dev[k * per].data_ptr() for k in range(K)]
    with pytest.raises(capi.SdrError) as ei:
        bank.graph_launch(ptrs)
    assert ei.value.code == capi.ERR_STATE
    bank.graph_release()
    bank.graph_capture(per)  # now with them
    bank.set_find_peaks(False)
    with pytest.raises(capi.SdrError) as ei:
        bank.graph_launch(ptrs)
    assert ei.value.code == capi.ERR_STATE
    bank.set_find_peaks(True)
    text = ["" for _ in range(tones)]
    for rep in range(2):
        bank.graph_launch([dev[(rep * K + k) * per].data_ptr() for k in range(K)])
        for k in range(K):
            res = bank.poll(wait=True)
            a = (rep * K + k) * per
            _check_delivery(res, out, a, a + per, tones, text)
    bank.sync()
    for lid in range(tones):
        assert text[lid] == ref.text(lid)
    bank.graph_release()
    bank.close()


def test_graph_release_with_many_unpolled_replays(capi):
    """Round 3's review: more than GRAPH_PHASES * RING (24) undelivered batches at sdr_graph_release used to be parked out of
    order (the set index wraps) and delivery stalled with 'results ... not where they should be'.  Five replays (30
    batches) without a single poll, the release (which also gives the replays' buffer sets back), then every batch polled
    in order against the oracle; six eager batches; a second capture - the sets come back - and two more replays."""
    import torch

    n, rate, tones, per = 1024, 96000, 4, 120
    edge = synth.default_edge_width(n)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=per, max_listeners=tones, max_peaks=128)
    K = bank.graph_batches
    reps1, reps2 = 5, 2
    frames = (reps1 + 1 + reps2) * K * per
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=4242)
    ref = orc.Receiver(rate, n, edge)
    stream = torch.cuda.Stream()
    bank.set_stream(stream.cuda_stream)
    for b in bins:
        assert bank.attach(0, int(b)) == ref.attach(int(b))
    out = ref.process(iq)
    bank.enable_results(True)
    dev = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    text = ["" for _ in range(tones)]
    bank.graph_capture(per)
    batch = 0
    for rep in range(reps1):
        bank.graph_launch([dev[(batch + k) * per].data_ptr() for k in range(K)])
        batch += K
    assert bank.results_pending == reps1 * K
    bank.graph_release()
    for i in range(reps1 * K):  # oldest first, nothing lost, nothing out of order
        res = bank.poll(wait=True)
        assert res["batch_index"] == i
        _check_delivery(res, out, i * per, (i + 1) * per, tones, text)
    for k in range(K):  # the eager ring takes over (a capture needs a multiple of K batches behind it)
        bank.process_device(dev[batch * per].data_ptr(), per)
        res = bank.poll(wait=True)
        assert res["batch_index"] == batch
        _check_delivery(res, out, batch * per, (batch + 1) * per, tones, text)
        batch += 1
    bank.graph_capture(per)
    for rep in range(reps2):
        bank.graph_launch([dev[(batch + k) * per].data_ptr() for k in range(K)])
        for k in range(K):
            res = bank.poll(wait=True)
            assert res["batch_index"] == batch + k
            _check_delivery(res, out, (batch + k) * per, (batch + k + 1) * per, tones, text)
        batch += K
    bank.sync()
    assert bank.total_frames == frames
    for lid in range(tones):
        assert text[lid] == ref.text(lid) and len(text[lid]) > 0
    bank.close()


def test_poll_from_a_consumer_thread(capi):
    """sdr_poll on a thread of its own while the producer thread keeps processing (the reference's Reporter and
    TextProcessor run on goroutines of their own): every batch arrives exactly once, in order, nothing is dropped,
    and the text adds up to the oracle's."""
    import threading
    import time

    n, rate, tones, per, batches = 1024, 96000, 4, 100, 40
    frames = per * batches
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=909)
    edge = synth.default_edge_width(n)
    ref = orc.Receiver(rate, n, edge)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=per, max_listeners=tones, max_peaks=128)
    for b in bins:
        assert bank.attach(0, int(b)) == ref.attach(int(b))
    out = ref.process(iq)
    bank.enable_results(True)
    got, text, stop = [], ["" for _ in range(tones)], threading.Event()

    def consume():
        while True:
            res = bank.poll(wait=True)
            if res is None:
                if stop.is_set():
                    return
                time.sleep(1e-4)
                continue
            got.append((res["batch_index"], res["first_frame"], len(res["edges"]), res["runes_dropped"], res["edges_dropped"]))
            for r in res["listeners"]:
                text[int(r["listener"])] += "".join(chr(int(x)) for x in res["runes"][r["first_rune"]:r["first_rune"] + r["n_runes"]])

    t = threading.Thread(target=consume)
    t.start()
    for k in range(batches):
        assert bank.process_host(iq[k * per:(k + 1) * per]) == per
    bank.sync()
    while bank.results_pending:
        time.sleep(1e-3)
    stop.set()
    t.join(timeout=30)
    assert not t.is_alive()
    assert [g[0] for g in got] == list(range(batches)) and [g[1] for g in got] == [k * per for k in range(batches)]
    assert all(g[3] == 0 and g[4] == 0 for g in got)
    total_edges = sum(int(np.count_nonzero(np.diff(np.concatenate([[0], out["deb"][:, l].astype(np.int8)])))) for l in range(tones))
    assert sum(g[2] for g in got) == total_edges
    for lid in range(tones):
        assert text[lid] == ref.text(lid) and len(text[lid]) > 0
    bank.close()


def test_decoder_waves_with_holes_in_the_pool(capi):
    """The decoder kernel fills every lane of a wave: lanes without a signal of their own follow a live one and store
    nothing.  A pool with holes - detached slots inside a wave's group of four, a group with a single live slot, whole
    groups empty, a pool size that is no multiple of the group - must decode exactly like the oracle, slot by slot,
    and leave the dead slots' state alone."""
    n, rate, tones, frames = 1024, 96000, 11, 600
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=4711)
    edge = synth.default_edge_width(n)
    ref = orc.Receiver(rate, n, edge)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=200, max_listeners=19, max_peaks=64)
    ids, rids = [], []
    for b in bins:
        ids.append(bank.attach(0, int(b)))
        rids.append(ref.attach(int(b)))
    assert ids == rids == list(range(tones))
    dead = [1, 2, 4, 5, 6, 9]  # group 0 keeps slots 0 and 3, group 1 keeps 7, group 2 keeps 8 and 10; groups 3, 4 empty
    state_before = {}
    ref.process(iq[:200])
    bank.process_host(iq[:200])
    for k in dead:
        state_before[k] = bank.read_decoder_state(0, k).copy()
        bank.detach(0, k)
        ref.detach(k)
    alive = [k for k in range(tones) if k not in dead]
    for lo in (200, 400):
        out = ref.process(iq[lo:lo + 200])
        bank.process_host(iq[lo:lo + 200])
        for k in alive:
            assert np.array_equal(bank.read_keying_bits(0, k), out["deb"][:, k]), (lo, k)
    for k in alive:
        assert bank.read_text(0, k) == ref.text(k), k
        assert np.array_equal(bank.read_decoder_state(0, k), ref.decoder_state(k)), k
    assert any(len(ref.text(k)) > 0 for k in alive)
    bank.close()


@pytest.mark.parametrize("find_peaks", [False, True])
def test_short_batches_without_sync_keep_the_ring_sets_safe(capi, find_peaks):
    """Batches shorter than a cumulation, no listeners, results off - stages that launch nothing (find_peaks on a batch
    that completes no cumulation, gather / decode without listener slots) must still publish their stage events, or
    the FFT of batch i + RING overwrites psd while k_cumulate(i) reads it.  More batches than ring sets are enqueued
    without a sync; the cumulation that the LAST batch completes carries every earlier batch's contribution."""
    import torch

    n, rate, per, batches = 4096, 192000, 60, 25
    frames = per * batches
    iq, _, _ = synth.make_band(frames, rate, n, 6, seed=321)
    edge = synth.default_edge_width(n)
    ref = orc.Receiver(rate, n, edge)
    out = ref.process(iq)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=per, max_listeners=0, max_peaks=128, find_peaks=find_peaks)
    bank.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    for k in range(batches):
        bank.process_device(dev[k * per].data_ptr(), per)
    bank.sync()
    assert bank.last_batch_chunks == 1  # frames 1440..1499 complete the cumulation that ends at frame 1499
    last = len(out["cumulation"]) - 1
    assert out["peak_frames"][last] == frames - 1
    assert _bits_equal(bank.read_cumulation(0, 0), out["cumulation"][last])
    _assert_records_equal(bank.read_frame_records(0), out["frames"][frames - per:])
    if find_peaks:
        assert bank.read_peaks(0, 0)[0] == out["peaks"][last]
    bank.close()


def test_results_with_an_empty_listener_pool(capi):
    """A bank created with max_listeners == 0 has no slot array; with bulk delivery on, k_pack_listen must touch none
    (it still delivers the drop counters) and sdr_poll hands out peaks only."""
    n, rate, frames = 1024, 96000, 250
    iq, _, _ = synth.make_band(frames, rate, n, 5, seed=55)
    edge = synth.default_edge_width(n)
    ref = orc.Receiver(rate, n, edge)
    out = ref.process(iq)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=256, max_listeners=0, max_peaks=128)
    bank.enable_results(True)
    assert bank.process_host(iq) == frames
    res = bank.poll(wait=True)
    assert res["n_frames"] == frames and len(res["listeners"]) == 0 and len(res["edges"]) == 0
    assert res["runes_dropped"] == 0 and res["edges_dropped"] == 0
    assert len(res["chunks"]) == 2
    for ch in res["chunks"]:
        gc = list(out["peak_frames"]).index(int(ch["frame"]))
        got = [tuple(int(p[k]) if k != "signal_value" else float(p[k]) for k in
                     ("from", "to", "from_frequency", "to_frequency", "signal_frequency", "signal_value", "signal_bin"))
               for p in res["peaks"][ch["first_peak"]:ch["first_peak"] + ch["n_peaks"]]]
        assert got == out["peaks"][gc] and len(got) >= 5
    bank.close()


def test_peak_frequencies_follow_the_batch_not_the_poll(capi):
    """sdr_set_center_frequency between a batch and its delivery: the batch's peaks keep the frequencies of ITS time
    (rx/receiver.go applies setters between frames; dsp/fft.go:95-135 maps bins with the mapping of that frame)."""
    n, rate, frames = 1024, 96000, 200
    iq, bins, _ = synth.make_band(frames, rate, n, 4, seed=91)
    edge = synth.default_edge_width(n)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=128, max_listeners=4, max_peaks=64)
    bank.enable_results(True)
    refs = []
    for k, cf in enumerate((7000000, 14000000)):
        bank.set_center_frequency(0, cf)
        bank.process_host(iq[k * 100:(k + 1) * 100])
        r = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=cf)
        if k:
            r.process(iq[:100])
        refs.append(r.process(iq[k * 100:(k + 1) * 100])["peaks"][0])
    bank.set_center_frequency(0, 21000000)  # after both batches, before either is polled
    for k in range(2):
        res = bank.poll(wait=True)
        ch = res["chunks"][0]
        got = [tuple(int(p[f]) if f != "signal_value" else float(p[f]) for f in
                     ("from", "to", "from_frequency", "to_frequency", "signal_frequency", "signal_value", "signal_bin"))
               for p in res["peaks"][ch["first_peak"]:ch["first_peak"] + ch["n_peaks"]]]
        assert got == refs[k] and len(got) >= 1
    bank.close()


def test_listeners_bound_inside_a_batch(capi):
    """sdr_defer_listen / sdr_poll_peaks / sdr_attach_at / sdr_process_listen (rx/receiver.go:404-426): a batch's
    spectral half runs first, listeners are bound to frames inside it afterwards - one at every cumulation boundary,
    as the reference binds them - and everything each listener produces (keying edges, runes) is what the oracle
    produces for a listener attached at that very frame, frame by frame.  The peaks of every cumulation are the
    oracle's as well, and the batch is delivered once, whole, by sdr_poll."""
    n, rate, tones = 1024, 96000, 6
    frames = 730
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=91)
    edge = synth.default_edge_width(n)
    # the oracle, frame by frame as the reference runs: a listener is attached after frames 99, 199, 299 and 499
    ref = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=3500000)
    lid0 = ref.attach(int(bins[0]))  # one listener is there from the start
    starts = {lid0: 0}
    outs = []
    pos = 0
    for boundary, b in zip((100, 200, 300, 500), bins[1:5]):
        outs.append((pos, ref.process(iq[pos:boundary])))
        pos = boundary
        starts[ref.attach(int(b))] = boundary
    outs.append((pos, ref.process(iq[pos:])))

    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=512, max_listeners=8, max_peaks=128)
    bank.set_center_frequency(0, 3500000)
    assert bank.attach(0, int(bins[0])) == lid0
    bank.enable_results(True)
    bank.defer_listen(True)
    text = {lid: "" for lid in starts}
    edges = {lid: [] for lid in starts}
    first = 0
    for a, e in ((0, 450), (450, 730)):
        assert bank.process_host(iq[a:e]) == e - a
        assert bank.listen_pending and bank.poll() is None  # the batch is not deliverable before its listen half
        with pytest.raises(Exception):
            bank.process_device(256, 10)  # refused until sdr_process_listen (the pointer is never touched)
        pk = bank.poll_peaks(wait=True)
        assert pk["first_frame"] == a and pk["n_frames"] == e - a and len(pk["listeners"]) == 0
        want_frames = [f for f in range(99, frames, 100) if a <= f < e]
        assert [int(c["frame"]) for c in pk["chunks"]] == want_frames
        # bind at every boundary of the batch, exactly where the oracle run attached
        for lid, s in sorted(starts.items()):
            if lid != lid0 and a < s <= e and s - 1 in want_frames:
                assert bank.attach_at(0, int(bins[lid]), s) == lid
        with pytest.raises(Exception):
            bank.attach_at(0, int(bins[5]), a - 1)  # not inside the waiting batch
        bank.process_listen()
        assert not bank.listen_pending
        res = bank.poll(wait=True)
        assert res["first_frame"] == a and res["n_frames"] == e - a
        assert [int(c["frame"]) for c in res["chunks"]] == want_frames
        for r in res["listeners"]:
            lid = int(r["listener"])
            ed = res["edges"][r["first_edge"]:r["first_edge"] + r["n_edges"]]
            edges[lid] += [(int(x["frame"]), int(x["state"])) for x in ed]
            text[lid] += "".join(chr(int(x)) for x in res["runes"][r["first_rune"]:r["first_rune"] + r["n_runes"]])
        first = e
    assert first == frames
    # oracle edges per listener from the segment outputs (a listener's column exists from its attach on)
    for lid, s in starts.items():
        want = []
        last = 0
        for base, out in outs:
            if out["deb"].shape[1] <= lid or base + out["deb"].shape[0] <= s:
                continue
            deb = out["deb"][:, lid].astype(np.int8)
            for j, v in enumerate(deb):
                if base + j >= s and v != last:
                    want.append((base + j, int(v)))
                    last = int(v)
        assert edges[lid] == want, lid
        assert text[lid] == ref.text(lid), lid
    assert sum(len(t) for t in text.values()) > 0
    bank.defer_listen(False)
    bank.close()


@pytest.mark.parametrize("mode", ["graph", "eager"])
def test_many_batches_with_an_erratic_consumer(capi, mode):
    """Eleven replays of six batches (the four groups of buffer sets graph mode rotates through are each used three
    times; eager: 66 batches over the six ring sets) while a consumer thread polls in fits and starts - so the producer
    finds unpolled batches in the sets it wants back and either yields to the consumer or parks them (host/delivery.h
    park_results).  Every batch must arrive once, in order, and be the oracle's, whichever way it went."""
    import random
    import threading
    import time

    import torch

    n, rate, tones, per = 1024, 96000, 3, 70
    edge = synth.default_edge_width(n)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=per, max_listeners=tones, max_peaks=128)
    K = bank.graph_batches
    replays = 11
    frames = replays * K * per
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=4711)
    ref = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=10100000)
    bank.set_center_frequency(0, 10100000)
    stream = torch.cuda.Stream()
    bank.set_stream(stream.cuda_stream)
    for b in bins:
        assert bank.attach(0, int(b)) == ref.attach(int(b))
    out = ref.process(iq)
    bank.enable_results(True)
    if mode == "graph":
        bank.graph_capture(per)
    dev = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    text = ["" for _ in range(tones)]
    got, errors, stop = [], [], threading.Event()
    rng = random.Random(5)

    def consume():
        try:
            while True:
                res = bank.poll(wait=rng.random() < 0.5)
                if res is None:
                    if stop.is_set() and bank.results_pending == 0:
                        return
                    time.sleep(rng.choice([0.0, 1e-4, 2e-3]))
                    continue
                a = int(res["batch_index"]) * per
                got.append(int(res["batch_index"]))
                _check_delivery(res, out, a, a + per, tones, text)
                if rng.random() < 0.3:
                    time.sleep(rng.choice([5e-4, 3e-3, 8e-3]))  # falls behind: the producer has to wait or park
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append(e)

    t = threading.Thread(target=consume)
    t.start()
    for rep in range(replays):
        if mode == "graph":
            bank.graph_launch([dev[(rep * K + k) * per].data_ptr() for k in range(K)])
        else:
            for k in range(K):
                bank.process_device(dev[(rep * K + k) * per].data_ptr(), per)
        if rep % 4 == 3:
            time.sleep(2e-3)
    bank.sync()
    stop.set()
    t.join(timeout=60)
    assert not t.is_alive() and not errors, errors
    assert got == list(range(replays * K))
    for lid in range(tones):
        assert text[lid] == ref.text(lid) and len(text[lid]) > 0
        assert np.array_equal(bank.read_decoder_state(0, lid), ref.decoder_state(lid))
    assert bank.read_drop_counters() == (0, 0)
    if mode == "graph":
        bank.graph_release()
    bank.close()


def test_decoder_scope_streams(capi):
    """cw.Decoder's own scope streams (cw/decode.go:228-243, :433-491): per tick the current run's duration, both
    adaptive thresholds with their low and high, and the state.  The device decodes in closed form between edges and
    keeps none of it; sdr_scope_read_decode replays the batch on the host from the decoder's state before it.  Against
    the oracle's literal Decoder ticked with the same keying, over two batches (the second replay starts mid-stream) and
    for a listener bound inside a batch (it ticks from its first frame on)."""
    n, rate, tones, per = 1024, 96000, 4, 260
    frames = 2 * per
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=1213)
    edge = synth.default_edge_width(n)
    ref = orc.Receiver(rate, n, edge)
    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=per, max_listeners=tones + 1, trace=True)
    for b in bins:
        assert bank.attach(0, int(b)) == ref.attach(int(b))
    out = ref.process(iq)
    decs = []
    for lid in range(tones):
        d = orc.Decoder(rate, n)
        d.reset()  # Listener.Attach -> demodulator.Reset (rx/listener.go:88)
        decs.append(d)

    def expected(d, keying, first):
        rows = []
        for j, st in enumerate(keying):
            d.tick(bool(st))
            s = d.state()  # ticks, onStart, offStart, wpm, on{low, high, last, thr}, off{low, high, last, thr}
            dur = s[0] - (s[1] if st else s[2])
            rows.append((first + j, dur, float(st), s[7], s[4], s[5], s[11], s[8], s[9]))
        return rows

    for k in range(2):
        a, e = k * per, (k + 1) * per
        assert bank.process_host(iq[a:e]) == per
        for lid in range(tones):
            fr = bank.scope_decode_frames(0, lid)
            want = expected(decs[lid], out["deb"][a:e, lid], a)
            assert len(fr) == per
            got = [tuple(float(fr[name][i]) if name != "frame" else int(fr[name][i]) for name in fr.dtype.names) for i in range(per)]
            assert got == want, f"batch {k} listener {lid}"
    bank.close()
    # a listener bound inside the batch: its decoder (fresh, Reset) takes its first tick at its first frame
    ref2 = orc.Receiver(rate, n, edge)
    late = capi.Bank(rate, n, edge_width=edge, max_batch_frames=per, max_listeners=2, trace=True)
    assert late.attach(0, int(bins[0])) == ref2.attach(int(bins[0]))
    o1 = ref2.process(iq[:130])
    lid1 = ref2.attach(int(bins[1]))
    o2 = ref2.process(iq[130:per])
    late.enable_results(True)
    late.defer_listen(True)
    assert late.process_host(iq[:per]) == per
    late.poll_peaks(wait=True)
    assert late.attach_at(0, int(bins[1]), 130) == lid1
    late.process_listen()
    late.poll(wait=True)
    fr = late.scope_decode_frames(0, lid1)
    d = orc.Decoder(rate, n)
    d.reset()
    want = expected(d, o2["deb"][:, lid1], 130)
    got = [tuple(float(fr[name][i]) if name != "frame" else int(fr[name][i]) for name in fr.dtype.names) for i in range(len(fr))]
    assert len(fr) == per - 130 and got == want
    fr0 = late.scope_decode_frames(0, 0)
    d0 = orc.Decoder(rate, n)
    d0.reset()
    want0 = expected(d0, np.concatenate([o1["deb"][:, 0], o2["deb"][:, 0]]), 0)
    assert [int(x) for x in fr0["frame"]] == list(range(per)) and [float(x) for x in fr0["duration"]] == [w[1] for w in want0]
    late.close()
    quiet = capi.Bank(rate, n, edge_width=edge, max_batch_frames=per, max_listeners=tones)
    quiet.attach(0, int(bins[0]))
    quiet.process_host(iq[:per])
    with pytest.raises(capi.SdrError) as ei:
        quiet.scope_decode_frames(0, 0)
    assert ei.value.code == capi.ERR_STATE
    quiet.close()
