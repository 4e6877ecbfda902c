"""sdrainer_amd — MI355X-native implementation of sdrainer's per-block IQ-strainer DSP.

The product is libsdrainer_hip.so (hand-written HIP for gfx950) behind the C ABI of
include/sdrainer_hip.h; this package holds its sources (csrc/), the ctypes binding the tests and the
benchmark use (capi), the synthetic IQ generator of the measurement plan (synth) and the band
sharding helpers for one-process-per-GPU runs (sharding).  There is no CPU fallback.
"""
__all__ = ["capi", "synth", "sharding"]
