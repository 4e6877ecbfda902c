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
python3 - $out <<'PY'
import re, sys
names = {"fb_prod": "production build", "fb_clock": "production + per-workgroup clock (SDR_FFT_CLOCK)", "fb_phases": "production + phase stamps of one workgroup",
         "fb_abl1": "no input read from LDS (constants instead; the LDS-DMA still runs)", "fb_abl3": "no twiddle loads",
         "fb_abl5": "no butterflies at all", "fb_abl6": "(almost) no psd stores", "fb_abl7": "no input, no stores",
         "fb_abl8": "input L2-resident (16 frames cycled)", "fb_abl10": "cross-wave exchange without its LDS traffic",
         "fb_abl11": "cross-wave exchange without its barriers", "fb_abl12": "cross-wave exchange without either",
         "fb_abl13": "no wave-local LDS exchange", "fb_abl14": "no register exchanges (layout B; layout A: unchanged kernel, slow outlier = different code)",
         "fb_abl15": "no input DMA at all (what a perfectly hidden input would leave)", "fb_abl16": "no input DMA, no stores"}
cur, rows = None, []
for line in open(sys.argv[1]):
    m = re.match(r"== (\S+) FPW=(\d+)", line)
    if m:
        cur = [m.group(1), m.group(2), None, None, None]
        rows.append(cur)
    m = re.match(r"single launch: min ([\d.]+) ms\s+median ([\d.]+) ms", line)
    if m and cur:
        cur[2], cur[3] = m.group(1), m.group(2)
    m = re.match(r"back to back: ([\d.]+) ms per launch", line)
    if m and cur:
        cur[4] = m.group(1)
print("%-10s %-3s %9s %9s %9s  %s" % ("binary", "fpw", "min ms", "median ms", "b2b ms", "what is removed (timing-only builds give wrong results by construction)"))
for r in rows:
    print("%-10s %-3s %9s %9s %9s  %s" % (r[0], r[1], r[2], r[3], r[4], names.get(r[0], "")))
PY
