"""The delivery state machine (sdrainer_amd/csrc/host/delivery.h: which finished batch sits in which buffer set's pinned
block or in the parked queue, who takes it, in what order) driven without a GPU: fake events, a fake device thread, a
producer that reuses ring sets and graph sets the way capi_process.hip / capi_graph.hip do, erratic / absent / blocking /
multiple consumers - under ThreadSanitizer.  The same header is what libsdrainer_hip.so runs behind sdr_poll."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "test_delivery_model.cpp")


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_delivery_model(tmp_path, sanitizer):
    exe = str(tmp_path / "test_delivery_model")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", "-pthread", f"-fsanitize={sanitizer}",
                         "-fno-sanitize-recover=all", "-o", exe, SRC], capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr and "error:" not in cc.stderr.replace("-Werror", ""):
        pytest.skip("this compiler has no -fsanitize=" + sanitizer)
    assert cc.returncode == 0, cc.stderr
    for _ in range(3):  # (the interleavings differ from run to run)
        run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0 and "ThreadSanitizer" not in run.stderr and "FAILED" not in run.stdout, run.stdout + run.stderr
    assert run.stdout.split() == ["erratic", "ok", "nosync", "ok", "blocking", "ok", "two", "ok", "graph", "ok", "re_enable", "ok"]
