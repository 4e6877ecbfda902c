#!/bin/bash
# Runs on the GPU box (from the repo root): everything tools/make_profiles.py condenses into profiles/ for a round.
#
# Reproducibility: the harness binaries are built BEFOREHAND on the CPU box by tools/build_tools.sh (flags in that
# script; every binary prints its flags and the sha256 of the kernel sources at the top of its output, and
# tools/bin/MANIFEST lists name / hash / flags).  This script REFUSES to run a binary whose recorded hash is not the hash
# of the sources it sits next to, and the production library must have been built from them too.
# rocprofv3: the program itself after `--`, PMC passes separate from the kernel trace, a timeout on every call.
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=${1:-r05}
PART=${2:-all}   # fft | pipeline | all (two gpurun calls fit the box limit more comfortably than one)
O=gpurun_out/prof_$R
[ "$PART" = pipeline ] || rm -rf $O; mkdir -p $O
HASH=$(python3 -c "from sdrainer_amd.csrc import build; print(build.source_hash())")
[ "$(cat sdrainer_amd/csrc/libsdrainer_hip.so.srchash 2>/dev/null)" = "$HASH" ] || { echo "libsdrainer_hip.so was not built from these sources"; exit 1; }
need() {  # binary names: each must be in the manifest with the current hash
  for n in "$@"; do
    grep -q "^$n $HASH" tools/bin/MANIFEST 2>/dev/null && [ -x tools/bin/$n ] || { echo "tools/bin/$n is missing or stale (tools/build_tools.sh)"; exit 1; }
  done
}
need fb_prod r32_prod r32_phases
cp tools/bin/MANIFEST $O/tool_manifest.txt
echo "sources sha256 $HASH" > $O/README.txt
set -e

if [ "$PART" != pipeline ]; then
# 1. the dominant kernel alone.  N = 16384 is served by k_fft_r32 (512 threads x 32 points); the 16-point-per-thread
#    k_fft_psd<14> (SDR_FFT_R32=0) and k_fft_psd<13> on the same number of samples stand beside it (verdict r4 1a).
{
  for fpw in 4 8 32; do echo "== r32_prod 8192 frames FPW=$fpw"; SDR_FFT_R32_FPW=$fpw SDR_R32_ONLY=1 timeout -k 5 90 tools/bin/r32_prod 8192 1; done
  echo "== r32_prod 2048 frames FPW=4"; SDR_R32_ONLY=1 timeout -k 5 90 tools/bin/r32_prod 2048 1
  echo "== r32_prod 8192 frames, word by word against k_fft_psd<14>"; timeout -k 5 120 tools/bin/r32_prod 8192 1
} > $O/r32_standalone.txt 2>&1
timeout -k 5 90 tools/bin/r32_phases 8192 1 > $O/r32_phases.txt 2>&1 || true
{
  echo "== k_fft_r32 through launch_fft: 8192 frames of N = 16384 (134 M samples)"; timeout -k 5 90 tools/bin/fb_prod 8192 14 1
  echo "== k_fft_psd<14> (SDR_FFT_R32=0): 8192 frames of N = 16384 (134 M samples)"; SDR_FFT_R32=0 SDR_FFT_FPW=1 timeout -k 5 90 tools/bin/fb_prod 8192 14 1
  echo "== k_fft_psd<13>: 16384 frames of N = 8192 (134 M samples)"; SDR_FFT_FPW=1 timeout -k 5 90 tools/bin/fb_prod 16384 13 1
  echo "== k_fft_psd<12>: 32768 frames of N = 4096 (134 M samples)"; SDR_FFT_FPW=1 timeout -k 5 90 tools/bin/fb_prod 32768 12 1
} > $O/fft_standalone_f8192.txt 2>&1
[ -x tools/ubench_f64 ] && timeout -k 5 120 tools/ubench_f64 > $O/ubench_f64.txt 2>&1 || true
[ -x tools/ubench_cvt ] && timeout -k 5 120 tools/ubench_cvt > $O/ubench_cvt.txt 2>&1 || true
echo "fft standalone done"

# 2. SQ / TCP counters of the standalone launch (one pass per group)
BIN=fb_prod OUT=$O/fft_sq_counters.txt tools/pmc_fft.sh > /dev/null 2>&1 || true

fi
[ "$PART" = fft ] && { echo "fft part done"; exit 0; }

# 3. the whole pipeline: kernel trace, HBM traffic (separate PMC passes), bench lines
# (the boxes of the pool differ by up to 7 %: the standalone FFT launch says which kind this one is, and the figure goes
# into the collection; MAX_PROBE_MS=<ms> gives a slow box back at once instead of spending the round's GPU budget on it)
PROBE=$(SDR_R32_ONLY=1 timeout -k 5 90 tools/bin/r32_prod 8192 1 | grep -o "back to back: [0-9.]*" | tail -1 | awk '{print $4}')
echo "box probe: k_fft_r32 standalone, 8192 frames, back to back: $PROBE ms per launch (this round's boxes: 0.502 ... 0.545)" | tee $O/box_probe.txt
if [ -n "$MAX_PROBE_MS" ] && [ "$(python3 -c "print(int(float('${PROBE:-9}') > float('$MAX_PROBE_MS')))")" = 1 ]; then echo "slow box: given back"; exit 7; fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/rocprof_bench.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_c5 -o t --output-format csv -- python3 bench.py --workload c5 --steps 150 --warmup 15 --no-cpu-baseline > $O/rocprof_bench_c5.log 2>&1
echo "kernel trace done"
# CU time per kernel (SQ_BUSY_CU_CYCLES, stages one after the other): what each stage holds against the FFT's whole CUs
bash tools/pmc_cutime.sh > $O/cu_time.txt 2>&1 || true
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --serial > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --serial > $O/pmc_write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch_c5 -o fetch --output-format csv -- python3 bench.py --workload c5 --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --serial > $O/pmc_fetch_c5.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $O/write_c5 -o write --output-format csv -- python3 bench.py --workload c5 --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --serial > $O/pmc_write_c5.log 2>&1
echo "pmc done"
python bench.py > $O/bench_full.json 2> $O/bench_full.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20_b.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20_c.json 2> /dev/null
python bench.py --frames 4096 --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench_f4096.json 2> /dev/null
python bench.py --frames 2048 --steps 2000 --warmup 200 --no-cpu-baseline > $O/bench_f2048.json 2> /dev/null
python bench.py --frames 2048 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_f2048_steps20.json 2> /dev/null
echo "bench done"
python bench.py --steps 100 --warmup 10 --kernel-breakdown --no-cpu-baseline --serial > $O/bench_serial.json 2> $O/bench_serial.err
python bench.py --steps 400 --warmup 40 --kernel-breakdown --no-cpu-baseline > $O/bench_insitu.json 2> $O/bench_insitu.err
python bench.py --workload c5 --steps 600 --warmup 60 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null
python bench.py --workload c2 --frames 4096 --steps 600 --warmup 60 --no-cpu-baseline > $O/bench_c2.json 2>/dev/null
python bench.py --workload c2 --frames 8192 --steps 600 --warmup 60 --no-cpu-baseline > $O/bench_c2_f8192.json 2>/dev/null
python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-delivery > $O/bench_nodelivery.json 2>/dev/null
# this round's two changes, each switched off: the 16-point FFT kernel, FindNoiseFloor's ordered chains
SDR_FFT_R32=0 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench_c3_r32off.json 2>/dev/null
SDR_NOISE_PATH=chains python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench_c3_chains.json 2>/dev/null
SDR_NOISE_PATH=chains python bench.py --workload c5 --steps 600 --warmup 60 --no-cpu-baseline > $O/bench_c5_chains.json 2>/dev/null
{
  echo "scan (default) against chains (SDR_NOISE_PATH=chains), 16-point FFT (SDR_FFT_R32=0) against k_fft_r32: value GS/s"
  for f in bench_full bench_c3_chains bench_c3_r32off bench_c5 bench_c5_chains; do python3 -c "import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('%-18s %8.2f GS/s  %.4f ms/step' % ('$f', d['value'], d['ms_per_step']))"; done
} > $O/noise_paths.txt 2>&1 || true
# the fences priced: the shipped (fenced) library against the program-order variant, same box, alternating
if [ -f sdrainer_amd/csrc/libsdrainer_hip_program_order.so ]; then
  {
    echo "product = fenced (SDR_SAFE_FENCES); variant = program order (build.py VARIANTS); alternating runs on one box"
    for i in 1 2 3; do
      for w in c3 c5; do
        a=$(python bench.py --workload $w --steps 600 --warmup 60 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
        b=$(SDR_HIP_LIB=$PWD/sdrainer_amd/csrc/libsdrainer_hip_program_order.so python bench.py --workload $w --steps 600 --warmup 60 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
        echo "$w run $i: fenced $a GS/s   program order $b GS/s"
      done
    done
  } > $O/fences.txt 2>&1 || true
fi
# hipGraph A/B: five fresh processes per workload (the stability claim is about fresh processes)
for i in 1 2 3 4 5; do
  python bench.py --workload c5 --steps 600 --warmup 60 --no-cpu-baseline --graph > $O/bench_graph_c5_$i.json 2>/dev/null || true
done
for i in 1 2 3; do
  python bench.py --steps 498 --warmup 48 --no-cpu-baseline --graph > $O/bench_graph_c3_$i.json 2>/dev/null || true
done
python tools/host_input_rate.py > $O/host_input_rate.txt 2>&1 || true
# (three fresh processes: the hunting phase is 11 ms long and one run in three has been seen to take twice that)
for i in 1 2 3; do timeout -k 10 300 python tools/strain_e2e.py 2>/dev/null | tail -1; done > $O/strain_e2e.json || true
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 || true
[ -f tools/abl/libntrace.so ] && { SDR_HIP_LIB=$PWD/tools/abl/libntrace.so timeout -k 10 120 python tools/noise_trace.py > $O/noise_trace.txt 2>&1; SDR_VAR_MFMA=0 SDR_HIP_LIB=$PWD/tools/abl/libntrace.so timeout -k 10 120 python tools/noise_trace.py >> $O/noise_trace.txt 2>&1; } || true
# k_listen_decode's stage clocks (tools/build_abl.sh decclk "-DSDR_DEC_CLOCK"): workgroup 0's chain waves (0, 2), character wave (1) and a helper (3)
[ -f tools/abl/libdecclk.so ] && { for w in c3 c2; do echo "== $w"; SDR_HIP_LIB=$PWD/tools/abl/libdecclk.so timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline --steps 6 --warmup 2 --serial 2>/dev/null | grep "decode clocks" | tail -4; done > $O/decode_clocks.txt; } || true
echo "all done"
tail -c 600 $O/bench_full.json
