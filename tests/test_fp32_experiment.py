"""The float32-FFT experiment behind DESIGN.md §3 stays reproducible: a complex64 FFT in front of the reference's
unchanged threshold logic misses north_star's 1e-5 magnitude tolerance on noise bins by an order of magnitude, while
flipping (almost) no keying decision - which is why the shipped FFT is float64 and why the reason is the tolerance,
not the decisions."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments"))


def test_fp32_fft_misses_the_magnitude_tolerance_not_the_decisions():
    import fp32_fft_decisions as exp

    r = exp.run(frames=200)
    # relative PSD error = twice the relative magnitude error
    assert 2e-5 < r["psd_rel_error_noise_bins"]["median"] < 1e-3
    assert not r["magnitude_tolerance_1e-5_met_on_noise_bins"]
    assert r["psd_rel_error_carrier_bins_key_down"]["median"] < 1e-6
    assert r["decisions"] == 200 * 256 and r["decisions_flipped"] <= 2
    assert r["cumulations"] == 2
