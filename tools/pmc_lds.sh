#!/bin/bash
# Development aid: LDS counters of the standalone k_fft_psd launch for a set of harness binaries (tools/bin/fb_*: the
# production kernel and timing-only builds with one LDS user removed), with and without the tap: which access pattern
# the bank conflicts belong to.   usage: tools/pmc_lds.sh [binary ...]
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SDR_TOOL_SHORT=1 SDR_FFT_FPW=1
bins="$@"; [ -z "$bins" ] && bins="fb_prod fb_abl1 fb_abl10 fb_abl13"
for b in $bins; do for tap in 0 256; do
  rm -rf gpurun_out/pmc_lds
  SDR_TAP=$tap timeout -k 10 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d gpurun_out/pmc_lds -o g --output-format csv -- tools/bin/$b 2048 > gpurun_out/pmc_lds.log 2>&1
  python3 - $b $tap <<'PY'
import csv, collections, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_lds/**/g_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fft_psd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print(f"{sys.argv[1]:10s} tap {sys.argv[2]:>3s}: conflict cycles {m.get('SQ_LDS_BANK_CONFLICT', -1):10.0f}  LDS active {m.get('SQ_LDS_IDX_ACTIVE', -1):10.0f}  LDS instructions {m.get('SQ_INSTS_LDS', -1):9.0f}")
PY
done; done
