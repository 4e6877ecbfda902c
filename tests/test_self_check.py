"""sdr_self_check: the one undocumented hardware behaviour the CHAIN kernels' results depend on - v_mfma_f64_4x4x4 with
B = 1 adds its four terms as a sequential, individually rounded chain (k_noise.hip variance_consumer_mfma;
dsp/fft.go:244-249) - is probed on the device the library runs on, by sdr_create of a bank that will use those kernels
(SDR_NOISE_PATH=chains; the default noise path, k_noise_scan.hip, does not touch the matrix pipe and is not probed), and
such a bank is refused where it does not hold.  Here: the check
passes on this device, and it is a real check - against the two other evaluation orders a matrix pipe might have (a
pairwise tree, the accumulator added last) it fails and sdr_create refuses the bank."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_check_passes_on_this_device():
    from sdrainer_amd import capi

    L = capi.load()
    assert L.sdr_self_check(0) == 0, L.sdr_last_error().decode()
    assert L.sdr_self_check(99) != 0  # no such device


@pytest.mark.parametrize("order", [1, 2])
def test_self_check_rejects_other_evaluation_orders(order):
    code = (
        "from sdrainer_amd import capi\n"
        "L = capi.load()\n"
        "rc = L.sdr_self_check(0)\n"
        "msg = L.sdr_last_error().decode()\n"
        "assert rc == capi.ERR_HIP and 'self-check failed' in msg, (rc, msg)\n"
        "try:\n"
        "    capi.Bank(192000, 4096, max_batch_frames=64, max_listeners=4)\n"
        "except capi.SdrError as e:\n"
        "    assert 'self-check failed' in str(e), str(e)\n"
        "    print('refused')\n"
    )
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SDR_SELF_CHECK_ORDER=str(order), SDR_NOISE_PATH="chains"), cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0 and "refused" in p.stdout, p.stdout + p.stderr[-2000:]


def test_default_path_is_not_probed():
    """With the default noise path a bank is created whatever the matrix pipe does (it is not used)."""
    code = ("from sdrainer_amd import capi\n"
            "b = capi.Bank(192000, 4096, max_batch_frames=64, max_listeners=4)\n"
            "b.close()\nprint('created')\n")
    env = {k: v for k, v in os.environ.items() if k != "SDR_NOISE_PATH"}
    p = subprocess.run([sys.executable, "-c", code], env=dict(env, SDR_SELF_CHECK_ORDER="1"), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "created" in p.stdout, p.stdout + p.stderr[-2000:]
