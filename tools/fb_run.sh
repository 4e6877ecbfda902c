#!/bin/bash
# Development aid: run a list of tools/bin/fb_* binaries (standalone FFT harness) on the GPU box, interleaved twice, and
# print one line per run.  usage: tools/fb_run.sh outdir name...   (SDR_TOOL_SHORT=1: timing + hash only)
out=$1; shift
mkdir -p $out
for rep in 1 2; do
  for n in "$@"; do
    SDR_TOOL_SHORT=1 timeout -k 10 60 tools/bin/$n ${FB_ARGS:-} > $out/$n.$rep.txt 2>&1 || { echo "$n FAILED"; tail -3 $out/$n.$rep.txt; exit 1; }
    printf "%-14s %s | %s | %s\n" $n "$(grep -o 'psd hash [0-9a-f]*' $out/$n.$rep.txt)" "$(grep -o 'min [0-9.]* ms' $out/$n.$rep.txt)" "$(grep -o 'back to back: [0-9.]* ms' $out/$n.$rep.txt)"
  done
done
