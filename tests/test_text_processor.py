"""rx.TextProcessor (rx/text_processor.go): the oracle restatement against the reference's own tests
(rx/text_processor_test.go), and the C++ host mirror against the oracle on scripted and random text."""
import json
import os
import random
import re
import subprocess

import pytest

from oracle import text_oracle as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- oracle vs rx/text_processor_test.go ---------------------------------------------------------
@pytest.mark.parametrize("preset,text,expected,expected_n,invalid", [
    ("", "", "", 0, False),
    ("", "abc", "abc", 3, False),
    ("123", "abc", "123abc", 3, False),
    ("1234567", "abcdef", "1234567abc", 3, False),
    ("1234567890", "abcdef", "1234567890", 0, True),
])
def test_text_window_write(preset, text, expected, expected_n, invalid):  # :10-69
    w = T.TextWindow(10)
    w.window[w.current] = preset.encode()
    n, err = w.write(text.encode())
    assert err == invalid
    assert n == expected_n
    assert w.string() == expected


def test_text_window_shift():  # :71-105
    w = T.TextWindow(10)
    w.shift()
    assert (w.current, w.string()) == (1, "")
    w.write(b"1234")
    w.shift()
    assert (w.current, w.string()) == (0, "1234")
    w.write(b"123456")
    w.shift()
    assert (w.current, w.string()) == (1, "23456")
    w.write(b"abcdefg")
    w.shift()
    assert (w.current, w.string()) == (0, "abcde")
    w.write(b"fg")
    w.shift()
    assert (w.current, w.string()) == (1, "cdefg")
    w.reset()
    assert (w.current, w.string()) == (0, "")


def test_text_window_find_next():  # :107-135
    w = T.TextWindow(10)
    a = re.compile("a")
    assert w.find_next(a, True) is None and w.search_point == 0
    w.write(b"abc")
    assert w.find_next(a, True) is not None and w.search_point == 1
    assert w.find_next(a, True) is None and w.search_point == 1
    w.write(b"1234567")
    w.shift()
    assert w.search_point == 0 and w.string() == "34567"
    w.write(b"abc")
    assert w.find_next(a, True) is not None and w.search_point == 6
    w.shift()
    assert w.search_point == 3 and w.string() == "67abc"


def test_text_window_find_next_include_tail():  # :137-147
    w = T.TextWindow(10)
    abc = re.compile("abc")
    w.write(b"12345abc")
    assert w.find_next(abc, False) is None
    assert w.find_next(abc, True) == "abc"


def test_collect_callsign():  # :149-162
    p = T.TextProcessor()
    for c in "cq cq cq de dl1abc dl1abc dl1abc pse k":
        p.write(c.encode())
    assert p.collected["DL1ABC"][1] == 3
    # third hearing crosses spottingThreshold (:18,:305-319)
    assert p.events[-1] == ("spotted", "DL1ABC")


def test_write_timeout():  # :164-179
    p = T.TextProcessor()
    for c in "cq de dl1abc":
        p.write(c.encode())
    assert "DL1ABC" not in p.collected
    p.write_timeout()
    assert p.collected["DL1ABC"][1] == 1


def test_parse_callsign_forms():
    assert T.parse_callsign("dl1abc") == "DL1ABC"
    assert T.parse_callsign("ea8/dl1abc/p") == "EA8/DL1ABC/P"
    assert T.parse_callsign("9a1a") == "9A1A"
    assert T.parse_callsign("dl1abc/mm") == "DL1ABC/MM"
    assert T.parse_callsign("5nn") is None
    assert T.parse_callsign("abc") is None


# ---- C++ host mirror vs oracle -------------------------------------------------------------------
@pytest.fixture(scope="module")
def exe():
    from sdrainer_amd.csrc import build
    build.build()
    csrc = os.path.join(ROOT, "sdrainer_amd", "csrc")
    out = os.path.join(ROOT, "tests", "host", "test_rx_host")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-o", out,
                           os.path.join(ROOT, "tests", "host", "test_rx_host.cpp"), "-L" + csrc, "-lsdrainer_hip",
                           "-Wl,-rpath," + csrc])
    return out


def run_script(exe, script):
    out = subprocess.run([exe, "text"], input="\n".join(script) + "\n", capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    events = []
    for line in out.stdout.splitlines():
        f = line.split()
        events.append((f[0], f[1], int(f[2]), int(f[3])) if f[0] == "decoded" else (f[0], f[1]))
    return events


def oracle_script(script):
    now = [0.0]
    p = T.TextProcessor(now=lambda: now[0])
    for line in script:
        op, arg = line[0], line[2:]
        if op == "W":
            for c in arg:
                p.write(c.encode())
        elif op == "B":
            p.write(arg.encode())
        elif op == "A":
            now[0] += float(arg)
            p.check_write_timeout()
        elif op == "R":
            p.restart()
    return p.events


def test_mirror_reference_scenarios(exe):
    script = ["W cq cq cq de dl1abc dl1abc dl1abc pse k"]
    got = run_script(exe, script)
    assert got == oracle_script(script)
    assert got == [("decoded", "DL1ABC", 1, 0), ("decoded", "DL1ABC", 2, 0), ("decoded", "DL1ABC", 3, 0),
                   ("spotted", "DL1ABC")]
    script = ["W cq de dl1abc", "A 4", "A 2"]  # 4 s: no time-out yet; 6 s > defaultWriteTimeout: tail is collected
    got = run_script(exe, script)
    assert got == oracle_script(script) == [("decoded", "DL1ABC", 1, 0)]


def test_mirror_spot_change_and_restart(exe):
    script = ["W  dl1abc dl1abc dl1abc de w1aw w1aw w1aw w1aw k", "R", "W  tu5nn tu5nn ea8/dj2xyz/p ea8/dj2xyz/p ea8/dj2xyz/p  "]
    got = run_script(exe, script)
    assert got == oracle_script(script)
    assert ("timeout", "DL1ABC") in got and ("spotted", "W1AW") in got
    # rune-by-rune writes accept a match as soon as one more byte follows it, so the "/p" that arrives
    # after "ea8/dj2xyz" is never part of the candidate (reference behaviour, :383-401 with includeTail=false)
    # Restart keeps the stale searchPoint (Reset, :348-353, does not clear it), which here swallows the first copy
    assert got[-1] == ("decoded", "EA8/DJ2XYZ", 2, 0)
    assert not any("TU5NN" in e[1] for e in got)


def test_mirror_random_text(exe):
    rng = random.Random(20250223)
    calls = ["dl1abc", "w1aw", "9a1a", "ea8/dj2xyz/p", "k3lr", "g4abc/mm", "ja1zzz", "5b4aa", "pa0x", "tu5nnx"]
    words = ["cq", "de", "test", "pse", "k", "tu", "5nn", "599", "73", "r", "=", "?", "agn", "bk", "qrz"]
    for trial in range(40):
        script = []
        for _ in range(rng.randrange(3, 12)):
            toks = []
            for _ in range(rng.randrange(1, 25)):
                r = rng.random()
                if r < 0.45:
                    c = rng.choice(calls)
                    if rng.random() < 0.15:  # a garbled copy
                        i = rng.randrange(len(c))
                        c = c[:i] + rng.choice("abcdefghijklmnopqrstuvwxyz0123456789/ ") + c[i + 1:]
                    toks.append(c)
                elif r < 0.9:
                    toks.append(rng.choice(words))
                else:
                    toks.append("".join(rng.choice("abcxyz0189/ ") for _ in range(rng.randrange(1, 30))))
            text = " ".join(toks) + (" " if rng.random() < 0.5 else "")
            script.append(("W " if rng.random() < 0.8 else "B ") + text)
            r = rng.random()
            if r < 0.3:
                script.append("A %d" % rng.randrange(1, 9))
            elif r < 0.35:
                script.append("R")
        assert run_script(exe, script) == oracle_script(script), json.dumps(script)


def test_mirror_on_recorded_stream_text(exe):
    """The text the decoder produces for the reference's recorded streams (tests/golden/cw_streams), fed on."""
    with open(os.path.join(ROOT, "tests", "golden", "cw_streams", "expected.json")) as f:
        expected = json.load(f)
    texts = []

    def walk(v):
        if isinstance(v, str):
            texts.append(v)
        elif isinstance(v, dict):
            for x in v.values():
                walk(x)
        elif isinstance(v, list):
            for x in v:
                walk(x)
    walk(expected)
    texts = [t for t in texts if t and all(32 <= ord(c) < 127 for c in t)]
    assert texts
    for t in texts:
        script = ["W " + t.lower(), "A 6"]
        assert run_script(exe, script) == oracle_script(script)
