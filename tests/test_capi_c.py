"""The drop-in boundary from PLAIN C (tests/host/test_capi_c.c): what the C half of a cgo shim sees.

CPU: include/sdrainer_hip.h compiles as C11 under -Wall -Werror -pedantic, the program links against the library, and
every struct has the size and the field offsets the ctypes binding (sdrainer_amd/capi.py) declares.
GPU: the same program runs create -> push_iq -> process_staged -> attach -> push / process -> poll -> destroy on a keyed
band; the peaks of every completed cumulation, every listener's keying edges and the decoded text it prints equal the
oracle's (the listeners are attached behind the first cumulation, as rx/receiver.go:409-426 binds them)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "test_capi_c.c")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from sdrainer_amd.csrc import build
    lib = build.build()
    out = str(tmp_path_factory.mktemp("capi_c") / "test_capi_c")
    libdir = os.path.dirname(lib)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-O1", "-o", out, SRC, "-L" + libdir,
                           "-l:" + os.path.basename(lib), "-Wl,-rpath," + libdir])
    return out


def _ctypes_layout(name, struct):
    lines = [f"sizeof {name} {C.sizeof(struct)}"]
    for field in struct._fields_:
        if field[0].startswith("reserved") or field[0] == "pad":
            continue
        lines.append(f"{name}.{field[0].rstrip('_')} {getattr(struct, field[0]).offset}")  # (`from` is a Python keyword: from_)
    return lines


def test_header_is_c_and_layouts_match_the_binding(exe):
    from sdrainer_amd import capi
    out = subprocess.run([exe, "layout"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    got = dict(line.rsplit(" ", 1) for line in out.stdout.strip().splitlines())
    assert got["abi"] == "2"
    for name, struct in (("sdr_config", capi.Config), ("sdr_peak", capi.Peak), ("sdr_results", capi.Results)):
        for line in _ctypes_layout(name, struct):
            key, val = line.rsplit(" ", 1)
            assert got[key] == val, (key, got.get(key), val)
    for name, dt in (("sdr_frame_rec", capi.FRAME_REC_DTYPE), ("sdr_edge", capi.EDGE_DTYPE), ("sdr_chunk_result", capi.CHUNK_RESULT_DTYPE),
                     ("sdr_listener_result", capi.LISTENER_RESULT_DTYPE)):
        assert got[f"sizeof {name}"] == str(dt.itemsize)
        for fname in dt.names:
            if fname in ("pad", "reserved"):
                continue
            assert got[f"{name}.{fname}"] == str(dt.fields[fname][1]), (name, fname)


@pytest.mark.gpu
def test_plain_c_caller_end_to_end(exe, tmp_path):
    from oracle import oracle as orc
    from sdrainer_amd import synth

    n, rate, tones, frames = 1024, 96000, 4, 700
    edge = synth.default_edge_width(n)
    iq, bins, _ = synth.make_band(frames, rate, n, tones, seed=4242)
    path = str(tmp_path / "iq.f32")
    np.ascontiguousarray(iq, dtype=np.float32).tofile(path)
    ref = orc.Receiver(rate, n, edge, 15.0, 1, center_frequency=0)
    out0 = ref.process(iq[:100])
    for b in bins:
        ref.attach(int(b))
    out1 = ref.process(iq[100:])
    p = subprocess.run([exe, "run", path, str(rate), str(n), str(frames), str(edge)] + [str(int(b)) for b in bins], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    assert lines[-1] == f"total {frames}"
    assert [ln for ln in lines if ln.startswith("attached")] == [f"attached {int(b)} -> {i}" for i, b in enumerate(bins)]
    assert all(ln == "dropped 0 0" for ln in lines if ln.startswith("dropped"))
    # peaks: every completed cumulation, in order, the oracle's tuples (the float32 value by its bits)
    want = []
    for out, base in ((out0, 0), (out1, 100)):
        for frame, peaks in zip(out["peak_frames"], out["peaks"]):
            want.append(f"chunk frame {int(frame) + base} found {len(peaks)}")
            for (frm, to, ff, tf, sf, val, sbin) in peaks:
                want.append(f"peak {frm} {to} {sbin} {ff} {tf} {sf} {np.float32(val).view(np.uint32):08x}")
    got = [ln for ln in lines if ln.startswith(("chunk", "peak"))]
    assert got == want
    # keying edges and text of every listener (second batch: frames 100 ..)
    deb = out1["deb"].astype(np.int8)
    for lid in range(tones):
        trans = np.flatnonzero(np.diff(np.concatenate([[0], deb[:, lid]])) != 0)
        edges = " ".join(f"{int(t) + 100}:{int(deb[t, lid])}" for t in trans)
        got_edges = [ln for ln in lines if ln.startswith(f"listener {lid} edges")]
        assert got_edges == [f"listener {lid} edges" + (" " + edges if edges else "")]
        runes = [ln for ln in lines if ln.startswith(f"listener {lid} runes")][0].split()[3:]
        assert "".join(chr(int(r)) for r in runes) == ref.text(lid) and len(runes) > 0
