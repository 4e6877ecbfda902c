#!/usr/bin/env python3
"""Counts the float64 vector arithmetic (v_add_f64 / v_mul_f64 / v_fma_f64 / v_fmac_f64) the compiled k_fft_psd
executes per IQ sample, for every block size, and writes profiles/f64_ops.json (read by bench.py for the
`roofline.compute` object).  The kernel is straight-line per frame, so the static count of the one-frame-per-
workgroup instantiation IS the dynamic count per thread; a thread handles R samples.

whole_path adds what the other kernels do per sample in float64:
  * dB projection (gomath.h db_fast_y + certificate: 5 fma + 3 add = 8 per evaluation).  Rounds 1-3 evaluated it for
    every bin of every frame (8 per sample); from round 4 on only for the bins FindPeaks looks at and the cumulation a
    batch leaves open (k_peaks.hip: about 6 % + 1 % of the bin-frames at config 3's 256 carriers and 8192-frame batches,
    every one of them for short batches, which take the exact kernel): 0.6 per sample at the benchmarked batch
  * FindNoiseFloor chains (k_noise.hip): one add per psd value inside the ten windows (0.73 of a frame at the
    default edge width) + sub, mul, add per value up to the winning window's end (on average 0.55 of that): ~1.9
(the rare literal-log fallbacks and the per-frame scalars are not counted).
Run on the CPU box: python tools/count_f64_ops.py
"""
import json
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sdrainer_amd", "csrc", "k_fft_psd.hip")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-DSDR_BUILD", "-mllvm", "-disable-machine-licm"]
DB_OPS, NOISE_OPS = 0.6, 1.9

with tempfile.TemporaryDirectory() as tmp:
    subprocess.check_call(["hipcc"] + FLAGS + ["-c", SRC, "-o", os.path.join(tmp, "k.o"), "--save-temps"], cwd=tmp,
                          stderr=subprocess.DEVNULL)
    asm = open(os.path.join(tmp, "k_fft_psd-hip-amdgcn-amd-amdhsa-gfx950.s")).read()

out = {}
for logn in range(9, 15):
    m = re.search(r"^(_ZN3sdr9k_fft_psdILi%dELb0E\w*):[^\n]*\n(.*?)s_endpgm" % logn, asm, re.S | re.M)
    body = m.group(2)
    counts = {op: len(re.findall(r"^\s+%s\b" % op, body, re.M)) for op in
              ("v_add_f64", "v_mul_f64", "v_fma_f64", "v_fmac_f64_e32")}
    valu = len(re.findall(r"^\s+v_", body, re.M))
    r = 16 if logn >= 10 else 8
    f64 = sum(counts.values())
    out[str(1 << logn)] = {"k_fft_psd": round(f64 / r, 2), "whole_path": round(f64 / r + DB_OPS + NOISE_OPS, 2),
                           "per_thread": counts, "valu_instructions_per_thread": valu, "samples_per_thread": r}
json.dump(out, open(os.path.join(ROOT, "profiles", "f64_ops.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
