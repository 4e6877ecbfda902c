#!/bin/bash
# Development aid: SQ counters of k_fft_psd over a short serial bench run, one rocprofv3 pass per
# counter group (no trace domains).  Output: gpurun_out/pmc_k1/<group>_counter_collection.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d gpurun_out/pmc_k1 -o g$i --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial > gpurun_out/pmc_k1_g$i.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/pmc_k1/**/g*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_fft_psd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
