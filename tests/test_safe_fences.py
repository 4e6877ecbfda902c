"""The places where the product relies on documented-by-measurement hardware ordering rather than on the memory model
- k_noise.hip's LDS flags ordered against data by program order only, k_fft_psd.hip's counted vmcnt - have a formally
fenced build (-DSDR_SAFE_FENCES: workgroup-scope fences, acquire / release flags, full waits), made by
__graft_entry__.build() beside the product.  The noise-floor chains, the thresholds that follow from them and the
multi-frame FFT workgroups must give the same bits under it: the parity tests that exercise them are run again, in a
process of their own, against that library."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_under_the_fenced_build():
    from sdrainer_amd.csrc import build

    lib = build.build_variant("safe_fences")
    assert os.path.exists(lib)
    env = dict(os.environ, SDR_HIP_LIB=lib, SDR_FFT_FPW="4")  # (multi-frame workgroups: the counted-wait path)
    sel = "test_receiver_run_bit_exact or test_nine_window_geometry or test_config5_geometry or test_batch_split_invariance_and_carry"
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", sel, "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout and "failed" not in p.stdout
    # the selection really ran against the variant
    q = subprocess.run([sys.executable, "-c", "from sdrainer_amd import capi; print(capi.LIB_PATH)"], env=env, cwd=ROOT,
                       capture_output=True, text=True)
    assert q.stdout.strip() == lib
