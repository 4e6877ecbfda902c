#!/bin/bash
# Development aid: SQ / TCP counters of the standalone k_fft_psd launch (tools/bin/$BIN, default fb_prod = the production
# kernel, built by tools/build_tools.sh; 2048 frames of N = 16384), one rocprofv3 pass per counter group, no trace
# domains, the program itself after `--`.
# Output: $OUT (default gpurun_out/pmc_fft.txt): the binary's header (flags, source hash) + per-dispatch averages.
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SDR_TOOL_SHORT=1 SDR_FFT_FPW=${SDR_FFT_FPW:-1} SDR_TAP=256
rm -rf gpurun_out/pmc_fft
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TOTAL_READ TCP_TOTAL_WRITE TCP_PENDING_STALL_CYCLES" \
           "TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TCR_TCP_STALL_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp -d gpurun_out/pmc_fft -o g$i --output-format csv -- tools/bin/${BIN:-fb_prod} 2048 > gpurun_out/pmc_fft_g$i.log 2>&1 || { echo "group $i failed: $grp"; tail -3 gpurun_out/pmc_fft_g$i.log; }
done
OUT=${OUT:-gpurun_out/pmc_fft.txt}
grep '^#' gpurun_out/pmc_fft_g1.log > $OUT
python3 - <<'PY' >> $OUT
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/pmc_fft/**/g*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_fft" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
cat $OUT
