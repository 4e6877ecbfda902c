#!/bin/bash
# Builds the standalone FFT timing / trace binaries used by tools/fft_matrix.sh (CPU box, cross-compiled).
# usage: build_fft_tools.sh name "extra flags" [name "extra flags" ...]
cd $(dirname $0)/..
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DSDR_BUILD -mllvm -disable-machine-licm -Iinclude"
mkdir -p tools/bin
while [ $# -gt 1 ]; do
  hipcc $F $2 -o tools/bin/$1 tools/fft_trace.hip 2>&1 | grep -E "error|spill" &
  shift 2
done
wait
ls tools/bin
