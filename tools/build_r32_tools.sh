#!/bin/bash
# Builds tools/fft_r32_bench.hip variants into tools/bin (cross-compiled on the CPU box):  name "flags" ...
cd $(dirname $0)/..
BASE="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DSDR_BUILD -DSDR_SAFE_FENCES -mllvm -disable-machine-licm -Iinclude -Wno-unused-lambda-capture"
HASH=$(python3 -c "from sdrainer_amd.csrc import build; print(build.source_hash())")
mkdir -p tools/bin
if [ $# -eq 0 ]; then set -- r32_prod "" r32_phases "-DSDR_R32_PHASES=100"; fi
pids=()
names=()
while [ $# -gt 1 ]; do
  name=$1; extra=$2; shift 2
  ( hipcc $BASE $extra "-DSDR_TOOL_FLAGS=\"$BASE $extra\"" "-DSDR_SRC_HASH=\"$HASH\"" -Rpass-analysis=kernel-resource-usage -o tools/bin/$name tools/fft_r32_bench.hip 2>&1 | grep -A9 "Function Name: .*k_fft_r32" | grep -E "error|VGPRs|Scratch" | sed "s/.*remark: */$name: /; s/\[-Rpass.*//" | tr '\n' ' '; echo ) &
  pids+=($!); names+=("$name")
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
done
for p in "${pids[@]}"; do wait $p; done
touch tools/bin/MANIFEST
for n in "${names[@]}"; do
  grep -v "^$n " tools/bin/MANIFEST > tools/bin/MANIFEST.tmp; mv tools/bin/MANIFEST.tmp tools/bin/MANIFEST
  echo "$n $HASH" >> tools/bin/MANIFEST
done
