#!/bin/bash
# Builds the standalone FFT harness binaries (tools/fft_bench.hip) into tools/bin, cross-compiled on the CPU box.
# Every binary carries the flags it was built with and the hash of the kernel sources (printed at the top of its output),
# and tools/bin/MANIFEST lists them: tools/run_profiles.sh refuses binaries whose hash is not the current sources'.
#   tools/build_tools.sh                      the standard set (production, clock, phases, the ablation matrix)
#   tools/build_tools.sh name "flags" ...     named variants
cd $(dirname $0)/..
BASE="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DSDR_BUILD -DSDR_SAFE_FENCES -mllvm -disable-machine-licm -Iinclude"
HASH=$(python3 -c "from sdrainer_amd.csrc import build; print(build.source_hash())")
mkdir -p tools/bin
if [ $# -eq 0 ]; then
  set -- fb_prod "" fb_clock "-DSDR_FFT_CLOCK" fb_phases "-DSDR_FFT_PHASES=1000" \
         fb_abl1 "-DSDR_ABLATE=1" fb_abl3 "-DSDR_ABLATE=3" fb_abl5 "-DSDR_ABLATE=5" fb_abl6 "-DSDR_ABLATE=6" fb_abl7 "-DSDR_ABLATE=7" \
         fb_abl8 "-DSDR_ABLATE=8" fb_abl10 "-DSDR_ABLATE=10" fb_abl11 "-DSDR_ABLATE=11" fb_abl12 "-DSDR_ABLATE=12" \
         fb_abl13 "-DSDR_ABLATE=13" fb_abl14 "-DSDR_ABLATE=14" fb_abl15 "-DSDR_ABLATE=15" fb_abl16 "-DSDR_ABLATE=16"
fi
pids=()
names=()
while [ $# -gt 1 ]; do
  name=$1; extra=$2; shift 2
  ( hipcc $BASE $extra "-DSDR_TOOL_FLAGS=\"$BASE $extra\"" "-DSDR_SRC_HASH=\"$HASH\"" -o tools/bin/$name tools/fft_bench.hip 2>&1 | grep -E "error|warning: .*spill" ; exit ${PIPESTATUS[0]} ) &
  pids+=($!); names+=("$name|$extra")
  # at most 6 compilers at a time (8 cores, 64 GiB)
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
touch tools/bin/MANIFEST
for nv in "${names[@]}"; do
  n=${nv%%|*}
  grep -v "^$n " tools/bin/MANIFEST > tools/bin/MANIFEST.tmp; mv tools/bin/MANIFEST.tmp tools/bin/MANIFEST
  echo "$n $HASH ${nv#*|}" >> tools/bin/MANIFEST
done
# (entries of binaries that are gone go too)
while read n rest; do [ -x tools/bin/$n ] && echo "$n $rest"; done < tools/bin/MANIFEST | sort > tools/bin/MANIFEST.tmp; mv tools/bin/MANIFEST.tmp tools/bin/MANIFEST
cat tools/bin/MANIFEST
exit $rc
