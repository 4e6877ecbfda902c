#!/bin/bash
# Development aid: standalone k_fft_psd timings (binaries from tools/build_fft_tools.sh).
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/fft_matrix.txt
: > $out
run() { # bin fpw frames [tap]
  echo "== $1 FPW=$2 frames=$3 tap=${4:-256}" >> $out
  SDR_TAP=${4:-256} SDR_TRACE_QUIET=${QUIET:-1} SDR_FFT_FPW=$2 timeout -k 5 60 tools/bin/$1 $3 >> $out 2>&1 || exit 1
}
for fpw in 1 2 4 8 16; do run ft_time1 $fpw 2048; done
run ft_time1 8 2048 0
run ft_time1 1 2048 0
run ft_clock1 8 2048
run ft_clock1 1 2048
grep -E "^==|single|launches" $out | paste - - - - | awk '{print $2,$3,$4,$5,"| single",$11,"| 1str",$19,"| 2str",$27}'
grep -A8 "ft_clock1" $out | grep -E "==|clock|span|lifetime|gap|XCC"
