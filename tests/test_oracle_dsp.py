"""Oracle checks for the dsp/ rows of SURVEY.md §8a.

What the reference pins (and is mirrored here): binToSpectrumIndex
(dsp/fft_test.go:10-29), FrequencyMapping (dsp/fft_test.go:31-50), Goertzel
blocksize law (dsp/dsp_test.go:151-161), PeaksTable (rx/peaks_test.go).
What it does NOT pin (parity unpinned): FFT values, PSD, MagnitudeIndB,
FindNoiseFloor, FindPeaks — those are cross-checked against numpy float64 and
against literal Python re-readings of the Go loops.
"""
import math

import numpy as np
import pytest

from oracle import oracle as orc


@pytest.mark.parametrize("bin_,expected", [(0, 256), (1, 257), (255, 511), (256, 0), (257, 1), (511, 255)])
def test_bin_to_spectrum_index(bin_, expected):
    assert orc.lib().orc_bin_to_spectrum_index(bin_, 512) == expected  # dsp/fft_test.go:16-21


@pytest.mark.parametrize("bin_,center", [(0, 7020000 - 24000), (256, 7020000)])
def test_frequency_mapping(bin_, center):
    L = orc.lib()  # dsp/fft_test.go:31-50
    assert L.orc_frequency_to_bin(48000, 512, 7020000, center) == bin_
    assert L.orc_bin_to_frequency(48000, 512, 7020000, bin_, 0.0) == center


def test_go_log10_matches_libm():
    rng = np.random.default_rng(1)
    xs = np.concatenate([10.0 ** rng.uniform(-30, 30, 20000), [1.0, 2.0, 0.5, 10.0, 100.0, 1e-300, 1e300]])
    got = np.array([orc.go_log10(float(x)) for x in xs])
    ref = np.log10(xs)
    # Go computes log2 as Log(frac)*(1/Ln2)+exp, which cancels near x ~ 1: the error is bounded in
    # absolute terms (a few 1e-17), not in ulps of a tiny result — so bound it by ulp(max(|ref|, 1))
    ulp = np.spacing(np.maximum(np.abs(ref), 1.0))
    assert np.max(np.abs(got - ref) / ulp) <= 2.0
    assert orc.go_log10(0.0) == -math.inf
    assert math.isnan(orc.go_log10(-1.0))
    assert orc.go_log10(1.0) == 0.0 and orc.lib().orc_go_log2(8.0) == 3.0


def test_go_sincos_matches_libm():
    rng = np.random.default_rng(2)
    xs = np.concatenate([rng.uniform(-2 * np.pi, 2 * np.pi, 20000), [0.0, -np.pi / 2, -np.pi, -np.pi / 4]])
    worst = 0.0
    for x in xs:
        s, c = orc.go_sincos(float(x))
        worst = max(worst, abs(s - math.sin(x)), abs(c - math.cos(x)))
    assert worst <= 2.3e-16  # within ~1 ulp at |value| <= 1


@pytest.mark.parametrize("n", [4, 8, 512, 4096, 16384])
def test_radix2_factors(n):
    w = orc.radix2_factors(n)
    k = np.arange(n)
    ref = np.exp(-2j * np.pi * k / n)
    assert np.max(np.abs(w - ref)) < 3e-16
    assert w[0] == 1 and w[n // 4] == -1j and w[n // 2] == -1 and w[3 * n // 4] == 1j  # literal size-4 table


@pytest.mark.parametrize("n", [2, 8, 512, 4096, 8192, 16384])
def test_fft_against_numpy(n):
    rng = np.random.default_rng(n)
    iq = rng.standard_normal(2 * n).astype(np.float32)
    got = orc.iq_fft(iq)
    x = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    ref = np.fft.fft(x)
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))


def test_fft_single_tone_is_exact_bin():
    n = 512
    t = np.arange(n)
    b = 37
    x = 0.1 * np.exp(2j * np.pi * b * t / n)
    iq = np.empty(2 * n, np.float32)
    iq[0::2], iq[1::2] = x.real, x.imag
    y = orc.iq_fft(iq)
    assert np.argmax(np.abs(y)) == b
    assert abs(abs(y[b]) - 0.1 * n) < 1e-4


def _projection_py(y, n):
    """Literal re-reading of dsp/fft.go:32-36,71-81 + receiver.go:376-378 with numpy scalars."""
    sp = np.empty(n, np.float32)
    psd = np.empty(n, np.float32)
    for i in range(n):
        k = (i + n // 2) % n
        p = np.float32(y[i].real * y[i].real + y[i].imag * y[i].imag)
        psd[k] = p
        db = np.float32(10.0 * orc.go_log10(20.0 * float(p) / float(n) ** 2)) if p > 0 else np.float32(-np.inf)
        sp[k] = np.float32(db + np.float32(120))
    return sp, psd


def test_projection_literal():
    n = 512
    rng = np.random.default_rng(7)
    iq = (1e-3 * rng.standard_normal(2 * n)).astype(np.float32)
    sp, psd = orc.iq_to_spectrum_and_psd(iq)
    sp2, psd2 = _projection_py(orc.iq_fft(iq), n)
    assert np.array_equal(psd, psd2) and np.array_equal(sp, sp2)
    # and against libm log10 within float32 rounding (<= 1 ulp of f32)
    ref = (10 * np.log10(20.0 * psd.astype(np.float64) / n ** 2)).astype(np.float32) + np.float32(120)
    assert np.max(np.abs(sp - ref)) <= np.spacing(np.float32(64.0))


def _find_noise_floor_py(psd, edge):
    """Literal re-reading of dsp/fft.go:215-252."""
    n = len(psd)
    window = (n - 2 * edge) // 10
    min_value = float(psd[0])
    s = 0.0
    count = 0
    first = True
    from_ = 0
    result_mean = 0.0
    result_from = result_to = 0
    for i in range(edge, n - edge):
        if count == 0:
            from_ = i
        if count == window:
            count = 0
            mean = s / float(window)
            if mean < min_value or first:
                min_value, first, result_mean, result_from, result_to = mean, False, mean, from_, i
            s = 0.0
        s += float(psd[i])
        count += 1
    s = 0.0
    for i in range(result_from, result_to + 1):
        d = float(psd[i]) - result_mean
        s += d * d
    return np.float32(min_value), s / float(window), result_from, result_to


@pytest.mark.parametrize("n,edge", [(512, 70), (512, 0), (4096, 560), (512, 100)])
def test_find_noise_floor_literal(n, edge):
    rng = np.random.default_rng(n + edge)
    psd = rng.exponential(1.0, n).astype(np.float32)
    psd[n // 3] = 1e6  # a tone before most windows: shows up in the quirky variance
    m, v = orc.find_noise_floor(psd, edge)
    m2, v2, rf, rt = _find_noise_floor_py(psd, edge)
    assert m == m2 and v == v2
    assert rf == edge  # resultFrom is always edgeWidth


def test_find_noise_floor_degenerate_window_is_nan():
    # N - 2*edge < 10 -> windowSize 0 -> NaN (SURVEY §7 hard parts); the C-ABI must reject such configs
    psd = np.ones(512, np.float32)
    m, v = orc.find_noise_floor(psd, 252)
    assert math.isnan(v)


def _find_peaks_py(cum, size, thr):
    peaks, cur = [], None
    thr = np.float32(thr)
    for i, v in enumerate(cum):
        value = np.float32(v) / np.float32(size)
        if cur is None and value > thr:
            cur = [i, i, value, i]
        elif cur is not None and value <= thr:
            cur[1] = i - 1
            peaks.append(tuple(cur))
            cur = None
        elif cur is not None and cur[2] < value:
            cur[2], cur[3] = value, i
    if cur is not None:
        cur[1] = len(cum) - 1
        peaks.append(tuple(cur))
    return peaks


def test_find_peaks_literal():
    n = 512
    rng = np.random.default_rng(3)
    cum = (rng.uniform(3000, 4000, n)).astype(np.float32)
    for b, w in [(0, 2), (100, 1), (200, 5), (300, 3), (510, 2)]:  # runs at both array ends too
        cum[b:b + w] = 9000 + np.arange(w) * (1 if b != 300 else 0)  # 300: equal maxima -> first bin wins
    got = orc.find_peaks(cum, 60.0, 48000, 7020000)
    ref = _find_peaks_py(cum, 100, 60.0)
    assert [(p[0], p[1], np.float32(p[5]), p[6]) for p in got] == [(a, b, v, sb) for a, b, v, sb in ref]
    assert got[-1][1] == n - 1  # open run closed at N-1 (dsp/fft.go:276-282)
    assert [p for p in got if p[0] == 300][0][6] == 300
    # frequencies: BinToFrequency with BinFrom/BinTo and the quadratic correction
    bs = 48000 / n
    for p in got:
        assert p[2] == 7020000 - 24000 + int(p[0] * bs + bs * -0.5)
        assert p[3] == 7020000 - 24000 + int(p[1] * bs + bs * 0.5)


def test_peak_center_correction():
    sp = np.array([1, 2, 4, 3, 1], np.float32)
    L = orc.lib()
    import ctypes as C
    ptr = sp.ctypes.data_as(C.POINTER(C.c_float))
    assert L.orc_peak_center_correction(2, ptr, 5) == (3 - 2) / (2 * (2 * 4 - 2 - 3))
    assert L.orc_peak_center_correction(0, ptr, 5) == 0 and L.orc_peak_center_correction(4, ptr, 5) == 0


def test_rolling_mean_is_f32_running_sum():
    L = orc.lib()  # dsp/dsp.go:257-268; the first n-1 means are biased low (App. C8)
    h = L.orc_rolling_mean_new(60)
    rng = np.random.default_rng(5)
    xs = rng.uniform(30, 50, 300).astype(np.float32)
    ring = np.zeros(60, np.float32)
    s = np.float32(0)
    nxt = 0
    for x in xs:
        s = np.float32(s - ring[nxt])
        ring[nxt] = x
        s = np.float32(s + x)
        nxt = (nxt + 1) % 60
        assert L.orc_rolling_mean_put(h, x) == np.float32(s / np.float32(60))
    L.orc_rolling_mean_free(h)


def test_goertzel_blocksize_law():
    L = orc.lib()  # dsp/dsp_test.go:151-161 and SURVEY §3.4
    assert L.orc_goertzel_blocksize(700.0, 48000, 0.005) == 207
    for f in range(301, 24000, 7):
        bs = L.orc_goertzel_blocksize(float(f), 48000, 0.005)
        assert abs(bs / 48000 - 0.005) <= 0.0017
    a = orc.AudioDemodulator(700.0, 48000)
    assert a.blocksize == 207 and abs(a.coeff - 1.99171368506) < 1e-11


def _tone(sr, f, n, amp=1.0):
    t = np.arange(n) / sr
    return (amp * np.cos(2 * np.pi * f * t)).astype(np.float32)


def test_goertzel_signal_state_properties():
    # dsp/dsp_test.go:25-149 (property tests): on pitch detected; half pitch, silence, DC not
    sr = 48000
    a = orc.AudioDemodulator(700.0, sr)
    a.set_scale(1.0)
    mags, raw, _ = a.write(_tone(sr, 700.0, a.blocksize * 10))
    assert raw.all()
    for sig in (_tone(sr, 350.0, 2070), np.zeros(2070, np.float32), np.ones(2070, np.float32)):
        b = orc.AudioDemodulator(700.0, sr)
        b.set_scale(1.0)
        _, raw, _ = b.write(sig)
        assert not raw.any()


def test_audio_demodulator_decodes_keyed_tone():
    # BASELINE config 1 plumbing: 48 kHz mono, 700 Hz tone keyed with Morse at 20 WPM -> text
    sr = 48000
    a = orc.AudioDemodulator(700.0, sr)
    a.set_scale(0.0)  # autoscale path, cw/audio.go:184-193
    bs = a.blocksize
    keying = orc.generate_stream(sr, bs, 20, "cq de dl1abc")
    env = np.repeat(keying, bs).astype(np.float32)
    sig = 0.8 * _tone(sr, 700.0, env.size) * env
    a.write(sig)
    a.close()
    assert a.text() == "cq de dl1abc"


# ---- rx/peaks_test.go ---------------------------------------------------------------------

def test_peaks_table_put_into_empty():
    t = orc.PeaksTable(512)
    e = t.put(234, 235)
    assert t.at(234) == e and t.at(235) == e and t.state(e) == t.NEW


def test_peaks_table_put_overlap_rules():
    t = orc.PeaksTable(12)  # rx/peaks_test.go:28-72
    p1 = t.place(3, 4, t.NEW)
    p2 = t.place(5, 6, t.NEW)
    p3 = t.place(8, 8, t.ACTIVE)
    p4 = t.place(10, 10, t.INACTIVE)
    n1, n2, n3, n4 = t.put(1, 2), t.put(4, 5), t.put(7, 8), t.put(10, 11)
    assert n3 == -1 and n4 == -1
    assert [t.at(i) for i in range(12)] == [-1, n1, n1, -1, n2, n2, -1, -1, p3, -1, p4, -1]
    assert p1 != n2 and p2 != n2


def test_peaks_table_cleanup():
    t = orc.PeaksTable(512)  # rx/peaks_test.go:74-124
    t.set_now(1000.0)
    e = t.put(234, 235)
    t.cleanup()
    assert t.at(234) == e
    t.set_now(1000.0 + 121.0)
    t.cleanup()
    assert t.at(234) == -1 and t.at(235) == -1
    t2 = orc.PeaksTable(512)
    t2.set_now(0.0)
    e = t2.put(234, 235)
    t2.activate(234, 235)
    t2.set_now(121.0)
    t2.cleanup()
    assert t2.at(234) == e
    t2.deactivate(234, 235)
    t2.cleanup()
    assert t2.at(234) == -1


def test_peaks_table_find_next():
    t = orc.PeaksTable(512)  # rx/peaks_test.go:126-143
    e = t.put(234, 235)
    assert t.find_next() == e
    t.activate(234, 235)
    assert t.find_next() == -1
    t.deactivate(234, 235)
    assert t.find_next() == -1


def test_fast_db_path_is_certified(tmp_path):
    """The FFT kernel's shortcut for the dB projection (gomath.h psd_value_in_db_fast) only ever returns
    the float32 the literal Go algorithm returns: checked on the CPU over ~5M values."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "emu_log")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                           os.path.join(root, "tests", "emu", "emu_log.cpp")])
    out = subprocess.run([exe, "2000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches 0" in out.stdout


def test_staged_listener_chain_matches_literal_ticks(tmp_path):
    """k_listen_decode's stages (cw_stages.h: debouncer per 64-tick word, edge list, threshold chain per edge,
    classification per edge, character assembly per edge) against literal BoolDebouncer.Debounce + Decoder.Tick calls
    on the CPU, tick by tick: same debounced bits, edges, runes, rune frames, and the same debouncer and decoder state at
    the end of every batch - 600 streams (keyed with jitter, glitches, over-long marks, long silences, over-long
    characters, plain noise) cut into batches of random length, debounce thresholds 1 ... 200, listeners that start
    inside a batch."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "emu_stages")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-o", exe,
                           os.path.join(root, "tests", "emu", "emu_stages.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches 0" in out.stdout and " 0 runes" not in out.stdout and " 0 edges" not in out.stdout


def test_cumulation_bound_is_an_upper_bound(tmp_path):
    """k_cum_bound's claim (gomath.h cum_bound_*): from the top halves of the psd words alone, an upper bound of every
    term of a cumulation (every exponent, dense and random mantissas, zero / subnormal / infinite / NaN), of the ordered
    float32 sum of 100 of them on top of a carry, and of what FindPeaks decides from it - for every block size, with the
    literal Go arithmetic on the CPU."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "emu_cum_bound")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(root, "tests", "emu", "emu_cum_bound.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and " 0 violations" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("layout_b_from", [15, 14])
def test_fft_phase_functions_match_oracle_bit_for_bit(tmp_path, layout_b_from):
    """The register/LDS index math and per-pass twiddle layout of the FFT kernel (fft_f64.h), emulated
    thread by thread on the CPU, against the oracle's stage-by-stage radix-2 FFT for every block size: LDS exchanges
    (padded additive address maps, bank-conflict audit), register exchanges (permlane swaps, lane rotations), twiddle
    rows in thread order, input staging.  Both layouts: A (shipped, all sizes) and B (experimental, N = 16384:
    cross-wave exchange first, registers only behind it)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "emu_fft")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", f"-DSDR_FFT_LAYOUT_B_FROM={layout_b_from}", "-o", exe,
                           os.path.join(root, "tests", "emu", "emu_fft.cpp"), "-ldl"])
    out = subprocess.run([exe, orc.build()], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(": 0 mismatches") == 6
    assert ("layout B" in out.stdout) == (layout_b_from <= 14)


def test_noise_floor_certification_against_the_literal_algorithm(tmp_path):
    """noise_cert.h: FindNoiseFloor's consumed values (float32 minimum mean, the two rolling-mean inputs, the winning
    window) from order-free sums, accepted only where the brackets around what the reference's ordered sums can be round
    and compare one way.  On the CPU, against the literal loops: noise, carriers, constant rows, means planted on float32
    rounding boundaries, tied windows, zeros / subnormals / a huge value, infinities and NaNs, eight geometries - an accepted
    frame has the reference's bits, specials are never accepted, ordinary rows nearly always are."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "emu_noise_cert")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(root, "tests", "emu", "emu_noise_cert.cpp")])
    out = subprocess.run([exe, "200"], capture_output=True, text=True)
    assert out.returncode == 0 and "\n0 violations" in out.stdout, out.stdout + out.stderr


def test_fft_r32_plan_matches_oracle_bit_for_bit(tmp_path):
    """The 32-points-per-thread plan of the N = 16384 FFT (fft_r32.h: 512 threads, 5 + 5 + 4 stages, the cross-wave exchange
    behind pass 0, a wave-local one behind pass 1, the psd row through LDS), emulated thread by thread on the CPU against
    the oracle's stage-by-stage radix-2 FFT; every LDS map audited against the MI355X banking rules, the input loads and
    pass-2 twiddle rows for contiguity per wave instruction."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "emu_fft_r32")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-o", exe,
                           os.path.join(root, "tests", "emu", "emu_fft_r32.cpp"), "-ldl"])
    out = subprocess.run([exe, orc.build()], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(": 0 mismatches") == 2
    assert out.stdout.count("worst write 1-way, worst read 1-way") == 2


def test_kiwi_iq_bytes_decode():
    # kiwi/client.go:298-308: big-endian int16 / 32767 in float32, after a 17-byte header
    vals = np.array([0, 1, -1, 32767, -32768, 12345, -12345, 256, 255], np.int16)
    payload = bytes(range(17)) + vals.astype(">i2").tobytes()
    got = orc.decode_iq_message(payload)
    want = vals.astype(np.float32) / np.float32(32767)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert got[3] == 1.0 and got[4] < -1.0  # -32768/32767: the reference does not clamp
