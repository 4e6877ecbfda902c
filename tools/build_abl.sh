#!/bin/bash
# Builds an alternative libsdrainer_hip.so into tools/abl/lib<name>.so with extra compiler flags
# (selected at run time with SDR_HIP_LIB).  usage: build_abl.sh name "extra flags"
cd $(dirname $0)/..
name=$1; extra=$2
src=sdrainer_amd/csrc
out=tools/abl/obj_$name
mkdir -p $out
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -DSDR_BUILD -DSDR_SAFE_FENCES $extra"
for f in $(python3 -c "from sdrainer_amd.csrc import build; print(' '.join(s[:-4] for s in build.SOURCES))"); do
  x=""; case $f in k_fft_psd|k_fft_r32) x="-mllvm -disable-machine-licm -Wno-unused-lambda-capture";; esac
  hipcc $F $x -c $src/$f.hip -o $out/$f.o 2>&1 | grep -E "error" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/abl/lib$name.so $out/*.o && ls -la tools/abl/lib$name.so
