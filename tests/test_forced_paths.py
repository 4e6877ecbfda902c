"""Two stages have two implementations and the launch picks by batch geometry: the cumulation (every slot exact for short
batches; an upper bound of every completed cumulation + the exact value only where FindPeaks looks, from 64 M samples per
batch on: k_peaks.hip) and FindNoiseFloor's variance chains (the float64 matrix pipe for short batches, two vector-ALU
chain groups per workgroup for long ones: k_noise.hip).  The parity tests run each at the sizes that pick it; here each is
FORCED onto the sizes that would not (SDR_CUM_BOUND / SDR_VAR_MFMA), in a process of its own, so that carries, partial
slots, nine-window geometries, batch splits and the bench-size batches all go through both.

Round 5: FindNoiseFloor has two paths as well - the one-pass scan with values certified where they are consumed
(k_noise_scan.hip, noise_cert.h; the default) and the ordered chains of rounds 1-4 (SDR_NOISE_PATH=chains).  The chains run
the same parity tests here; and the scan is run with its literal fallback FORCED for every frame and for every third one
(SDR_NOISE_FORCE_EXACT), so that the list of flagged frames, the exact kernel behind it and the mix of both are exercised
(a frame is flagged by itself about three times in 10^5)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ("test_receiver_run_bit_exact or test_nine_window_geometry or test_config5_geometry or test_batch_split_invariance_and_carry"
         " or test_scan_segment_geometries")


@pytest.mark.parametrize("env, files, sel", [
    ({"SDR_CUM_BOUND": "1", "SDR_VAR_MFMA": "0"}, ["tests/test_gpu_parity.py", "tests/test_dsp_golden.py"], SMALL + " or golden"),
    ({"SDR_CUM_BOUND": "0", "SDR_VAR_MFMA": "1"}, ["tests/test_gpu_parity_bench_sizes.py"], "config3 or config5 or config2"),
    ({"SDR_NOISE_PATH": "chains"}, ["tests/test_gpu_parity.py", "tests/test_dsp_golden.py", "tests/test_gpu_parity_bench_sizes.py"],
     SMALL + " or golden or config3 or config2"),
    ({"SDR_NOISE_PATH": "chains", "SDR_CUM_BOUND": "1", "SDR_VAR_MFMA": "0"}, ["tests/test_gpu_parity.py"], SMALL),
    ({"SDR_NOISE_FORCE_EXACT": "1"}, ["tests/test_gpu_parity.py", "tests/test_dsp_golden.py"], SMALL + " or golden"),
    ({"SDR_NOISE_FORCE_EXACT": "3", "SDR_CUM_BOUND": "1"}, ["tests/test_gpu_parity.py", "tests/test_gpu_parity_bench_sizes.py"],
     SMALL + " or config5 or config2"),
    # the randomised streams and cuts of the fuzz test through the cumulation's bound-and-refine path and the wide
    # refinement workgroups (its batches are far too short to pick either by themselves: round 4's advice)
    # k_fft_r32 and its wide tap on batches far too short to pick them, under the bound-and-refine path: listeners'
    # columns from the wide tap, a signal without a listener from the psd array, both refinement shapes
    ({"SDR_FFT_R32": "1", "SDR_CUM_BOUND": "1", "SDR_REFINE_WIDE": "1"}, ["tests/test_gpu_parity.py"], "test_scan_segment_geometries"),
    ({"SDR_FFT_R32": "1", "SDR_CUM_BOUND": "1", "SDR_REFINE_WIDE": "0"}, ["tests/test_gpu_parity.py"], "test_scan_segment_geometries"),
    ({"SDR_FFT_R32": "1", "SDR_CUM_BOUND": "1", "SDR_FUZZ_N": "16384", "SDR_FUZZ_SEEDS": "6"}, ["tests/test_gpu_fuzz.py"], "random_streams"),
    # config 3's 256 listeners x 2048 frames with the bound forced on: the refinement reads 768 columns per cumulation from
    # the wide tap, and the peaks are the oracle's
    ({"SDR_CUM_BOUND": "1"}, ["tests/test_gpu_parity_bench_sizes.py"], "config3"),
    ({"SDR_CUM_BOUND": "1", "SDR_REFINE_WIDE": "1"}, ["tests/test_gpu_fuzz.py"], "random_streams"),
    ({"SDR_CUM_BOUND": "1", "SDR_REFINE_WIDE": "0", "SDR_NOISE_PATH": "chains", "SDR_VAR_MFMA": "0"}, ["tests/test_gpu_fuzz.py"], "random_streams"),
])
def test_parity_with_the_other_implementation_forced(env, files, sel):
    p = subprocess.run([sys.executable, "-m", "pytest", *[os.path.join(ROOT, f) for f in files], "-q", "-x", "-m", "gpu", "-k", sel,
                        "-p", "no:cacheprovider"], env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout and "failed" not in p.stdout
