#!/bin/bash
# Development aid: registers / spills / LDS of every kernel in one .hip source, as the compiler reports them.
# usage: tools/kernel_resources.sh sdrainer_amd/csrc/k_fft_psd.hip [extra flags]
src=$1; shift
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DSDR_BUILD -Iinclude"
case $src in *k_fft_psd*|*fft_bench*) F="$F -mllvm -disable-machine-licm";; esac
/opt/rocm/bin/hipcc $F "$@" --cuda-device-only -Rpass-analysis=kernel-resource-usage -c $src -o /dev/null 2>&1 | python3 -c '
import re,sys
cur=None
for l in sys.stdin:
    m=re.search(r"remark:\s+(.*?) \[-Rpass",l)
    if not m:
        if "error" in l: print(l.rstrip())
        continue
    t=m.group(1)
    if t.startswith("Function Name:"):
        cur={"name":t.split(": ")[1]}
    elif cur is not None:
        k,v=t.split(":",1); cur[k.strip()]=v.strip()
        if k.strip().startswith("LDS Size"):
            print("%-90s vgpr %4s agpr %3s sgpr %4s scratch %5s spill(v/s) %s/%s occ %s lds %s" % (cur["name"][:90],cur.get("VGPRs"),cur.get("AGPRs"),cur.get("TotalSGPRs"),cur.get("ScratchSize [bytes/lane]"),cur.get("VGPRs Spill"),cur.get("SGPRs Spill"),cur.get("Occupancy [waves/SIMD]"),cur.get("LDS Size [bytes/block]")))
'
