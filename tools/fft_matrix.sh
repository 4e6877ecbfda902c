#!/bin/bash
# Development aid: standalone k_fft_project timings (binaries from tools/build_fft_tools.sh).
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/fft_matrix.txt
: > $out
run() { # bin fpw stagger frames
  echo "== $1 FPW=$2 STAGGER_US=$3 frames=$4" >> $out
  SDR_TRACE_QUIET=${QUIET:-1} SDR_FFT_FPW=$2 SDR_FFT_STAGGER_US=$3 timeout -k 5 60 tools/bin/$1 $4 >> $out 2>&1 || exit 1
}
run ft_time0 1 0 2048
for b in ft_abl3 ft_abl4 ft_abl10 ft_abl11 ft_abl12 ft_abl13 ft_abl14 ft_abl15; do run $b 1 0 2048; done
run ft_clock 8 0 2048
run ft_clock 1 0 2048
grep -E "^==|single|launches|clock" $out | paste - - - - | awk '{print $2,$3,$4,$5,"| single",$11,"| 2str",$27}'
grep clock $out
