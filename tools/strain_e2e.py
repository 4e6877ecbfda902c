#!/usr/bin/env python3
"""End-to-end strain mode (rx::Receiver over the C ABI: discover -> attach -> decode -> spots) on device-resident
IQ, BASELINE config 3 with a 256-listener pool.  Prints MSamples/s while hunting and with the pool full.
Never the bench `value`: that is the resident kernel pipeline with a pre-attached pool (bench.py)."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "sdrainer_amd", "csrc")
LIB = os.path.join(ROOT, "tools", "bin", "libstrain_e2e.so")


def main():
    import torch

    from sdrainer_amd import capi, synth
    from sdrainer_amd.csrc import build

    build.build()
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", LIB, os.path.join(ROOT, "tools", "strain_e2e.cpp"),
                           "-L" + CSRC, "-lsdrainer_hip", "-Wl,-rpath," + CSRC])
    capi.load()
    lib = C.CDLL(LIB)
    lib.strain_e2e.restype = C.c_int
    lib.strain_e2e.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.POINTER(C.c_double)]
    rate, n, tones, frames = 2_000_000, 16384, 256, 2048
    pool = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    iq, bins, _ = synth.make_band_torch(frames, rate, n, tones, seed=3017, device="cuda", free_last_window=True)
    torch.cuda.synchronize()
    out = (C.c_double * 8)()
    # the round-2 way first (every 100-frame segment resolved on the host before the next), on a short run
    os.environ["SDR_RX_NO_SPECULATION"] = "1"
    rc = lib.strain_e2e(C.c_void_p(iq.data_ptr()), frames, rate, n, pool, 2048, frames, out)
    assert rc == 0, rc
    classic = list(out)
    del os.environ["SDR_RX_NO_SPECULATION"]
    rc = lib.strain_e2e(C.c_void_p(iq.data_ptr()), frames, rate, n, pool, 2048, 400 * frames, out)
    assert rc == 0, rc
    o = list(out)
    res = {
        "workload": "BASELINE config 3 through rx::Receiver (strain mode), device-resident IQ",
        "pool": pool, "listeners_bound": int(o[4]),
        "hunting": {"frames": int(o[0]), "seconds": round(o[1], 4), "MSamples_per_s": round(o[0] * n / o[1] / 1e6, 1),
                    "segment_frames": "up to 2048: spectra first, the boundary decisions, then the listeners (sdr_attach_at)"},
        "hunting_one_cumulation_per_round_trip": {"frames": int(classic[0]), "seconds": round(classic[1], 4),
                                                  "MSamples_per_s": round(classic[0] * n / classic[1] / 1e6, 1), "segment_frames": 100,
                                                  "listeners_bound": int(classic[4])},
        "pool_full": {"frames": int(o[2]), "seconds": round(o[3], 4), "MSamples_per_s": round(o[2] * n / o[3] / 1e6, 1),
                      "segment_frames": 2048},
        "runes_decoded": int(o[5]), "callsigns_decoded": int(o[6]), "callsigns_spotted": int(o[7]),
    }
    print(json.dumps(res))


if __name__ == "__main__":
    main()
