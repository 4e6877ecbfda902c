"""The FMA experiment of DESIGN.md section 3 stays runnable: tests/experiments/fma_fft_decisions.py (the oracle's frame
loop with only the FFT and the psd contracted the way a compiler would fuse them) on a short run - psd words differ
rarely and by one ulp, no decision changes.  The 8192-frame numbers are in profiles/r04_fma_experiment.json."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "experiments"))


def test_fma_experiment_runs_and_changes_no_decision():
    import fma_fft_decisions

    r = fma_fft_decisions.run(frames=200)
    assert r["psd_words"] == 200 * 16384 and r["max_ulps_apart"] <= 1 and r["fraction"] < 1e-4
    assert r["decisions"] == 200 * 256 and r["decisions_flipped"] == 0 and r["keying_edges_changed"] == 0
    assert r["cumulations"] == 2 and r["peak_lists_that_differ_in_bins"] == 0
