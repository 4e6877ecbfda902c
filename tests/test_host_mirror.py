"""C++ host mirror of rx.Receiver / PeaksTable / ListenerPool (sdrainer_amd/csrc/host/rx.h) over the C ABI."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdrainer_amd", "csrc")
EXE = os.path.join(ROOT, "tests", "host", "test_rx_host")


@pytest.fixture(scope="module")
def exe():
    from sdrainer_amd.csrc import build
    build.build()
    src = os.path.join(ROOT, "tests", "host", "test_rx_host.cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-pthread", "-o", EXE, src, "-L" + CSRC, "-lsdrainer_hip",
                           "-Wl,-rpath," + CSRC])
    return EXE


def test_host_bookkeeping_matches_reference_tests(exe):
    # rx/peaks_test.go + rx/listener_test.go scenarios; no GPU call is made
    out = subprocess.run([exe, "cpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip() == "ok"


@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_host_bookkeeping_under_sanitizers(exe, tmp_path, sanitizer):
    """The same CPU scenarios (peaks table, listener pool, text window, the call-sign matchers against their regular
    expressions, the worker pool's hand-out) built with AddressSanitizer + UBSan and with ThreadSanitizer: clean.  (UBSan
    found the one defect this file has seen so far: the listener's clock lambda read a member that was initialised
    after the text processor whose constructor calls it.)"""
    src = os.path.join(ROOT, "tests", "host", "test_rx_host.cpp")
    out_exe = str(tmp_path / "test_rx_host_san")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-pthread", f"-fsanitize={sanitizer}", "-fno-sanitize-recover=all",
                         "-o", out_exe, src, "-L" + CSRC, "-lsdrainer_hip", "-Wl,-rpath," + CSRC], capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr:
        pytest.skip("this compiler has no -fsanitize=" + sanitizer)
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([out_exe, "cpu", "20000"], capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="halt_on_error=1"))
    assert run.returncode == 0 and run.stdout.strip() == "ok", run.stdout[-1000:] + run.stderr[-3000:]


@pytest.mark.gpu
def test_strain_mode_receiver_against_oracle(exe, tmp_path):
    """Strain mode end to end: peaks discovered every 100 frames, one new listener bound per cumulation
    (rx/receiver.go:409-426) — compared with the same policy simulated on the CPU oracle."""
    from oracle import oracle as orc
    from sdrainer_amd import synth

    rate, n, frames, pool, tones = 48000, 512, 950, 4, 6
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=77)
    path = tmp_path / "iq.f32"
    iq.astype(np.float32).tofile(path)
    out = subprocess.run([exe, "strain", str(path), str(rate), str(n), str(frames), str(pool)], capture_output=True,
                         text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = json.loads(out.stdout)
    assert got["frames"] == frames

    # oracle-driven simulation of the reference's strain loop with the deterministic FindNext (linear scan)
    ref = orc.Receiver(rate, n, 70, 15.0, 1, center_frequency=7020000)
    table = {}  # bin -> state ('new' | 'active')
    attached = []
    for c in range(frames // 100 + 1):
        chunk = iq[100 * c:100 * (c + 1)]
        if len(chunk) == 0:
            break
        hunting = len(attached) < pool
        ref.set_find_peaks(hunting)
        res = ref.process(chunk)
        if not hunting or res["n_chunks"] == 0:
            continue
        for p in res["peaks"][0]:
            sb = p[6]
            if table.get(sb) != "active":  # Put refuses to overlap an active peak, replaces a new one
                table[sb] = ("new", p)
        new = sorted(b for b, v in table.items() if v != "active" and v[0] == "new")
        if new:
            sb = new[0]
            p = table[sb][1]
            table[sb] = "active"
            lid = ref.attach(sb)
            attached.append((lid, sb, p[4]))
    assert len(got["listeners"]) == len(attached) == pool
    for (lid, sb, freq), l in zip(attached, got["listeners"]):
        assert l["bin"] == sb and l["frequency"] == freq
        text = bytes(ord(ch) for ch in l["text"]).decode("utf-8")
        assert text == ref.text(lid)
        assert len(text) > 0
    assert got["events"] == [f"+rx{i + 1}@{a[2]}" for i, a in enumerate(attached)]


@pytest.mark.gpu
def test_strain_mode_spots_callsigns(exe, tmp_path):
    """IQ in, spots out: the runes the GPU decoders emit go through rx::TextProcessor
    (rx/text_processor.go) and reach the Reporter as CallsignDecoded / CallsignSpotted with the listener's
    id and signal frequency (rx/listener.go:70-83).  Expected events: the text oracle fed with each
    listener's text rune by rune, as cw/decode.go:352 writes it."""
    from oracle import text_oracle
    from sdrainer_amd import synth

    rate, n, frames, pool, tones = 48000, 512, 4200, 2, 3
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=5)
    path = tmp_path / "iq.f32"
    iq.astype(np.float32).tofile(path)
    out = subprocess.run([exe, "strain", str(path), str(rate), str(n), str(frames), str(pool)], capture_output=True,
                         text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = json.loads(out.stdout)
    assert got["frames"] == frames and len(got["listeners"]) == pool
    spotted = 0
    for l in got["listeners"]:
        text = bytes(ord(ch) for ch in l["text"]).decode("utf-8")
        p = text_oracle.TextProcessor()
        for ch in text:
            p.write(ch.encode("utf-8"))
        want = []
        for e in p.events:
            if e[0] == "decoded":
                want.append(f"{l['id']} decoded {e[1]} {l['frequency']} {e[2]} {e[3]}")
            else:
                want.append(f"{l['id']} {e[0]} {e[1]} {l['frequency']}")
        mine = [c for c in got["callsigns"] if c.startswith(l["id"] + " ")]
        assert mine == want
        spotted += any(c == f"{l['id']} spotted DL1ABC {l['frequency']}" for c in mine)
    assert spotted == pool


@pytest.mark.gpu
def test_strain_mode_strongest_first_many_listeners(exe, tmp_path):
    """strain mode with the strongest-new-peak-first policy is reproducible without a seed.
    C2 geometry (192 kS/s, N = 4096), 16 carriers of different strength, a pool of 16: one listener is bound
    per cumulation, strongest carrier first - compared with the same policy simulated on the CPU oracle."""
    from oracle import oracle as orc
    from sdrainer_amd import synth

    rate, n, pool, tones = 192000, 4096, 16, 16
    frames = 100 * (pool + 2) + 37
    iq, bins, key = synth.make_band(frames, rate, n, tones, seed=2024)
    # give every carrier its own level: scale carrier k by (1 + k/8) in the frequency domain (exact bins)
    spec = np.fft.fft(iq[:, 0::2].astype(np.float64) + 1j * iq[:, 1::2].astype(np.float64), axis=1)
    for k, b in enumerate(bins):
        spec[:, (int(b) + n // 2) % n] *= 1.0 + k / 8.0
    t = np.fft.ifft(spec, axis=1)
    iq = np.empty((frames, 2 * n), np.float32)
    iq[:, 0::2] = t.real
    iq[:, 1::2] = t.imag
    path = tmp_path / "iq.f32"
    iq.tofile(path)
    out = subprocess.run([exe, "strain", str(path), str(rate), str(n), str(frames), str(pool), "strongest"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = json.loads(out.stdout)
    assert got["frames"] == frames

    ref = orc.Receiver(rate, n, 70 * n // 512, 15.0, 1, center_frequency=7020000)
    table, attached = {}, []
    for c in range(frames // 100 + 1):
        chunk = iq[100 * c:100 * (c + 1)]
        if len(chunk) == 0:
            break
        hunting = len(attached) < pool
        ref.set_find_peaks(hunting)
        res = ref.process(chunk)
        if not hunting or res["n_chunks"] == 0:
            continue
        for p in res["peaks"][0]:
            sb = p[6]
            if table.get(sb) != "active":
                table[sb] = ("new", p)
        new = [(v[1][5], -b, b) for b, v in table.items() if v != "active" and v[0] == "new"]
        if new:
            _, _, sb = max(new)  # strongest signal value, ties: lowest bin
            p = table[sb][1]
            table[sb] = "active"
            attached.append((ref.attach(sb), sb, p[4]))
    assert len(attached) == pool == len(got["listeners"])
    # the swap-remove pool keeps binding order while nothing is released
    for (lid, sb, freq), l in zip(attached, got["listeners"]):
        assert l["bin"] == sb and l["frequency"] == freq
        assert bytes(ord(ch) for ch in l["text"]).decode("utf-8") == ref.text(lid)
    assert len({a[1] for a in attached}) == pool  # sixteen different carriers


def _simulate_strain(iq, rate, n, pool_size, silence, attachment, edge, center=7020000):
    """The reference's strain loop (rx/receiver.go:353-463) frame by frame on the CPU oracle, with the stream clock
    (frame f happens at (f + 1) * n / rate) and the deterministic FindNext (lowest new bin): time-outs are evaluated
    after every frame's Listen, exactly where the reference evaluates them (:396-400)."""
    from oracle import oracle as orc

    ref = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=center)
    ref.set_find_peaks(True)
    T = float(n) / float(rate)
    ids = [f"rx{pool_size - i}" for i in range(pool_size)]  # IDPool: Pop takes from the back
    pool, table, events, event_frames, sessions = [], {}, [], [], []
    for f in range(len(iq)):
        now = float(f + 1) * T
        res = ref.process(iq[f:f + 1])
        detached = []
        for l in pool:  # pool order
            text = ref.text(l["lid"])
            if len(text) > l["len"]:
                l["len"] = len(text)
                l["last_write"] = now
            if now - l["last_attach"] > attachment or now - l["last_write"] > silence:
                table[l["bin"]] = ("inactive", l["peak"])
                ref.detach(l["lid"])
                events.append(f"-{l['id']}@{l['peak'][4]}")
                event_frames.append(f + 1)
                detached.append(l)
        for l in detached:  # ListenerPool.Release: swap-remove, id back on the stack
            i = pool.index(l)
            ids.append(l["id"])
            if len(pool) > 1:
                pool[i] = pool[-1]
            pool.pop()
            l["text"] = ref.text(l["lid"])
            sessions.append(l)
        if (f + 1) % 100 == 0 and len(pool) < pool_size and res["n_chunks"]:
            for p in res["peaks"][0]:
                sb = p[6]
                if table.get(sb, ("none",))[0] in ("active", "inactive"):
                    continue
                table[sb] = ("new", p)
            new = sorted(b for b, v in table.items() if v[0] == "new")
            if new:
                sb = new[0]
                p = table[sb][1]
                table[sb] = ("active", p)
                l = {"id": ids.pop(), "lid": ref.attach(sb), "bin": sb, "peak": p, "last_attach": now, "last_write": now,
                     "len": 0}
                pool.append(l)
                events.append(f"+{l['id']}@{p[4]}")
                event_frames.append(f + 1)
    for l in pool:
        l["text"] = ref.text(l["lid"])
    return events, event_frames, pool, sessions


@pytest.mark.gpu
def test_strain_mode_finite_timeouts_per_frame(exe, tmp_path):
    """Listener time-outs fire at the frame the reference would fire them (rx/receiver.go:396-400,
    rx/listener.go:126-136), not at the next batch boundary: attachment time-outs in the middle of long batches
    with a full pool, silence time-outs after the signals stop, re-binding at the following cumulation boundary."""
    from sdrainer_amd import synth

    rate, n, pool, tones = 48000, 512, 2, 4
    frames = 1900
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=31)
    quiet, _, _ = synth.make_band(frames, rate, n, 0, seed=32)
    iq[1200:] = quiet[1200:]  # every signal stops at frame 1200: silence time-outs follow
    silence, attachment = 2.5, 6.3  # seconds = 234.4 and 590.6 frames of 10.67 ms
    path = tmp_path / "iq.f32"
    iq.astype(np.float32).tofile(path)
    out = subprocess.run([exe, "strain", str(path), str(rate), str(n), str(frames), str(pool), "linear", str(silence),
                          str(attachment), "256"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = json.loads(out.stdout)
    events, event_frames, live, sessions = _simulate_strain(iq, rate, n, pool, silence, attachment, 70)
    kinds = {e[0] for e in events}
    assert kinds == {"+", "-"} and len([e for e in events if e[0] == "-"]) >= 3, events
    assert got["events"] == events
    assert got["event_frames"] == event_frames
    assert [l["id"] for l in got["listeners"]] == [l["id"] for l in live]
    for g, l in zip(got["listeners"], live):
        assert g["bin"] == l["bin"] and bytes(ord(ch) for ch in g["text"]).decode("utf-8") == l["text"]


@pytest.mark.gpu
@pytest.mark.parametrize("silence,attachment", [(1e9, 1e9), (2.5, 6.3)])
def test_strain_mode_long_segments_decide_like_one_cumulation_at_a_time(exe, tmp_path, silence, attachment):
    """Discovery over long segments (rx.h discoverAhead: the spectral half of up to 1024 frames first, the decisions of
    every cumulation boundary in it next, the listeners - bound to frames INSIDE the segment, sdr_attach_at - last)
    against the per-frame simulation of rx/receiver.go:388-426 on the oracle, and against the same receiver made to
    resolve every 100-frame cumulation on the host before the next (SDR_RX_NO_SPECULATION=1): same listeners on the
    same peaks at the same frames, same time-outs, same text, same callsign events."""
    from sdrainer_amd import synth

    rate, n, pool, tones = 48000, 512, 5, 7
    frames = 2350
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=58)
    quiet, _, _ = synth.make_band(frames, rate, n, 0, seed=59)
    iq[1700:] = quiet[1700:]
    path = tmp_path / "iq.f32"
    iq.astype(np.float32).tofile(path)
    cmd = [exe, "strain", str(path), str(rate), str(n), str(frames), str(pool), "linear", str(silence), str(attachment),
           "1024", "1000"]
    ahead = subprocess.run(cmd, capture_output=True, text=True)
    assert ahead.returncode == 0, ahead.stdout + ahead.stderr
    one = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, SDR_RX_NO_SPECULATION="1"))
    assert one.returncode == 0, one.stdout + one.stderr
    got, ref = json.loads(ahead.stdout), json.loads(one.stdout)
    assert got["frames"] == frames
    assert got["events"] == ref["events"] and got["event_frames"] == ref["event_frames"]
    assert got["listeners"] == ref["listeners"]
    assert sorted(got["callsigns"]) == sorted(ref["callsigns"])  # (per listener in order; listeners interleave freely)
    for lid in {c.split(" ")[0] for c in got["callsigns"]}:
        assert [c for c in got["callsigns"] if c.startswith(lid + " ")] == [c for c in ref["callsigns"] if c.startswith(lid + " ")]
    events, event_frames, live, sessions = _simulate_strain(iq, rate, n, pool, silence, attachment, 70)
    assert len([e for e in events if e[0] == "+"]) >= pool
    assert got["events"] == events and got["event_frames"] == event_frames
    for g, l in zip(got["listeners"], live):
        assert g["bin"] == l["bin"] and bytes(ord(ch) for ch in g["text"]).decode("utf-8") == l["text"]


@pytest.mark.gpu
def test_strain_mode_discovery_ahead_at_wideband_geometry(exe, tmp_path):
    """The same two receivers - decisions made ahead over long segments, and one cumulation per host round trip - at a
    wideband geometry (2 MS/s, N = 16384, a pool of 24 filled strongest-first from 40 carriers, 2048-frame segments):
    same listeners on the same peaks at the same frames, same text, same callsign events."""
    from sdrainer_amd import synth

    rate, n, pool, tones, frames = 2_000_000, 16384, 24, 40, 2900
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=612, free_last_window=True)
    path = tmp_path / "iq.f32"
    iq.astype(np.float32).tofile(path)
    cmd = [exe, "strain", str(path), str(rate), str(n), str(frames), str(pool), "strongest", "1e9", "1e9", "2048", "2048"]
    ahead = subprocess.run(cmd, capture_output=True, text=True)
    assert ahead.returncode == 0, ahead.stdout[-2000:] + ahead.stderr[-2000:]
    one = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, SDR_RX_NO_SPECULATION="1"))
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    got, ref = json.loads(ahead.stdout), json.loads(one.stdout)
    assert got["frames"] == frames and len(got["listeners"]) == pool
    assert got["events"] == ref["events"] and got["event_frames"] == ref["event_frames"]
    assert len(got["events"]) == pool and all(f % 100 == 0 for f in got["event_frames"])  # bound at cumulation boundaries
    assert got["event_frames"][-1] <= 100 * (pool + 3)  # about one listener per cumulation
    assert got["listeners"] == ref["listeners"]
    assert sum(len(l["text"]) for l in got["listeners"]) > 0
    for lid in {c.split(" ")[0] for c in ref["callsigns"]}:
        assert [c for c in got["callsigns"] if c.startswith(lid + " ")] == [c for c in ref["callsigns"] if c.startswith(lid + " ")]


@pytest.mark.gpu
def test_decode_mode_vfo_listener(exe, tmp_path):
    """DecodeMode (rx/receiver.go:272-297): SetVFOOffset forces a peak at the VFO frequency, the receiver's single
    listener decodes it; retuning resets the pool of one and a fresh listener takes over.  No peak scan runs."""
    from oracle import oracle as orc
    from sdrainer_amd import synth

    rate, n, frames, tones = 48000, 512, 1400, 3
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=9)
    sb = int(bins[1])
    center = 7020000
    offset = int((sb - n // 2) * rate / n) + 20  # a frequency inside bin sb, relative to the centre
    path = tmp_path / "iq.f32"
    iq.astype(np.float32).tofile(path)
    out = subprocess.run([exe, "decode", str(path), str(rate), str(n), str(frames), str(offset)], capture_output=True,
                         text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = json.loads(out.stdout)
    assert got["frames"] == frames and got["bin"] == sb and got["peaks_found"] == 0
    half = frames // 2
    ref = orc.Receiver(rate, n, 70, 15.0, 1, center_frequency=center)
    a = ref.attach(sb)
    ref.process(iq[:half])
    ref.detach(a)
    b = ref.attach(sb)
    ref.process(iq[half:])
    assert bytes(ord(ch) for ch in got["text0"]).decode("utf-8") == ref.text(a) != ""
    assert bytes(ord(ch) for ch in got["text1"]).decode("utf-8") == ref.text(b) != ""
    f = center + offset
    assert got["events"] == [f"+rx1@{f}", f"-rx1@{f}", f"+rx1@{f}"]
