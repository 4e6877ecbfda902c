"""Builds libsdrainer_hip.so in-tree with hipcc for gfx950 (MI355X only).

-ffp-contract=off is part of the numerical contract: the kernels reproduce the Go reference's
rounding, and Go on amd64 never fuses a multiply-add (see gomath.h, fft_f64.h).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libsdrainer_hip.so")
SOURCES = ["k_fft_psd.hip", "k_fft_r32.hip", "k_noise.hip", "k_noise_scan.hip", "k_listen.hip", "k_peaks.hip", "k_unpack.hip", "k_results.hip", "capi_bank.hip", "capi_process.hip", "capi_results.hip", "capi_graph.hip", "capi_read.hip", "sdr_audio.hip"]
HEADERS = ["sdr_device.h", "bank.h", "host/delivery.h", "fft_f64.h", "fft_r32.h", "noise_cert.h", "gomath.h", "cw_decoder.h", "cw_stages.h", "twiddles.h", "host/frequency_mapping.h",
           "../../include/sdrainer_hip.h"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-DSDR_BUILD",
         # The formally fenced protocol wherever a kernel could lean on measured-but-undocumented ordering instead
         # (k_noise.hip's LDS flags, k_fft_psd.hip's counted vmcnt): round 5 priced it - config 3 200.2 / 198.4 GS/s, config
         # 5's share 233.3 / 232.8 without / with on the default path, 181.5 / 180.0 with the chain kernels - under a
         # percent, so the product is the fenced build.  The other one stays as a variant ("program_order").
         "-DSDR_SAFE_FENCES"]
# k_fft_psd: machine-LICM parks literal constants in VGPRs for the whole kernel; the FFT needs all 128
# registers a 1024-thread workgroup leaves it, and four parked constants are four spilled data registers.
EXTRA_FLAGS = {"k_fft_psd.hip": ["-mllvm", "-disable-machine-licm", "-Wno-unused-lambda-capture"],
               "k_fft_r32.hip": ["-mllvm", "-disable-machine-licm", "-Wno-unused-lambda-capture"]}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libsdrainer_hip.so cannot be built (no CPU fallback exists)")


STAMP = LIB + ".srchash"


def source_hash() -> str:
    """sha256 over every source, header, flag and this script: a library is reused only if it was built
    from exactly these bytes (mtimes do not survive a copy to another box)."""
    import hashlib

    h = hashlib.sha256()
    for d in [os.path.join(HERE, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def _drop_link_temporaries():
    # the link step leaves its offload-bundler temporaries beside the output (libsdrainer_hip.so.N.hipv4-... /
    # .host-...): they are not part of the product (and a build that was interrupted leaves them behind)
    import glob

    for tmp in glob.glob(LIB + ".*.hipv4-*") + glob.glob(LIB + ".*.host-*"):
        try:
            os.remove(tmp)
        except OSError:
            pass


def build(force: bool = False, verbose: bool = False) -> str:
    _drop_link_temporaries()
    if not force and not needs_build():
        return LIB
    cc = hipcc()
    objs = []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(HERE, s.replace(".hip", ".o"))
        cmd = [cc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    _drop_link_temporaries()
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return LIB


VARIANTS = {
    # without the workgroup-scope fences / with the counted vmcnt of rounds 2-4 (k_noise.hip, k_fft_psd.hip: ordering by
    # program order, a measured property of the hardware): same bits, under a percent faster; kept so that the price of the
    # fences stays measurable (tests/test_safe_fences.py runs the parity tests against it too)
    "program_order": ["-USDR_SAFE_FENCES"],
}


def variant_path(name: str) -> str:
    return os.path.join(HERE, f"libsdrainer_hip_{name}.so")


def build_variant(name: str, force: bool = False) -> str:
    """A diagnostic build of the whole library with extra flags, beside the product (selected with SDR_HIP_LIB)."""
    extra = VARIANTS[name]
    lib, stamp = variant_path(name), variant_path(name) + ".srchash"
    want = source_hash() + " " + " ".join(extra)
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return lib
    cc = hipcc()
    objdir = os.path.join(HERE, f"obj_{name}")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for s in SOURCES:
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        subprocess.check_call([cc] + FLAGS + extra + EXTRA_FLAGS.get(s, []) + ["-c", os.path.join(HERE, s), "-o", obj])
        objs.append(obj)
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    import glob

    for tmp in glob.glob(lib + ".*.hipv4-*") + glob.glob(lib + ".*.host-*"):
        os.remove(tmp)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--variants" in sys.argv:
        for v in VARIANTS:
            print(build_variant(v, force="--force" in sys.argv))
