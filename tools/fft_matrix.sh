#!/bin/bash
# Development aid: standalone k_fft_psd runs of the binaries tools/build_tools.sh made (all of them, or those named).
#   tools/fft_matrix.sh [out-file] [binary ...]     env FPWS="1 8" = SDR_FFT_FPW values to run each binary with
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=${1:-gpurun_out/fft_matrix.txt}; shift
mkdir -p $(dirname $out)
: > $out
bins="$@"
[ -z "$bins" ] && bins=$(ls tools/bin | grep '^fb_')
for b in $bins; do
  for fpw in ${FPWS:-1}; do
    echo "== $b FPW=$fpw" >> $out
    SDR_FFT_FPW=$fpw timeout -k 5 90 tools/bin/$b ${FRAMES:-2048} ${LOGN:-14} ${BANDS:-1} >> $out 2>&1 || { echo "FAILED $b" >> $out; exit 1; }
  done
done
grep -E "^==|psd hash|single launch|back to back" $out | paste - - - - | awk '{print $2, $3, "|", $6, "|", $12, $13, $14, $15, "| b2b", $23}'
