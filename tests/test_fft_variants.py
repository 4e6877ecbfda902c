"""The FFT kernel's multi-frame variant (SDR_FFT_FPW > 1: a workgroup takes consecutive frames, prefetches the next
one by LDS-DMA behind a counted vmcnt wait and taps frame f-1 near the end of frame f) is a run-time knob read once per
process, so it gets a process of its own: the golden digests and the tap-dependent parity tests must hold for it too."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("fpw", ["2", "8"])
def test_multi_frame_workgroups_bit_exact(fpw):
    env = dict(os.environ, SDR_FFT_FPW=fpw)
    # children start before this process touches the GPU (it never does)
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_dsp_golden.py",
                          "tests/test_gpu_parity.py", "-k",
                          "golden or spectrum_psd or receiver_run or batch_split or multi_band or config5 or full_size or graph"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout
