#!/bin/bash
# Runs on the GPU box (from the repo root): everything tools/make_profiles.py condenses into profiles/ for this round.
# rocprofv3: the program itself after `--`, PMC passes separate from the kernel trace, a timeout on every call.
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=${1:-r02}
O=gpurun_out/prof_$R
rm -rf $O && mkdir -p $O
set -e
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline > $O/rocprof_bench.log 2>&1
echo "kernel trace done"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --serial > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --serial > $O/pmc_write.log 2>&1
echo "pmc done"
python bench.py > $O/bench_full.json 2> $O/bench_full.err
echo "bench done"
python bench.py --steps 300 --warmup 30 --kernel-breakdown --no-cpu-baseline --serial > $O/bench_serial.json 2> $O/bench_serial.err
python bench.py --steps 1000 --warmup 100 --kernel-breakdown --no-cpu-baseline > $O/bench_insitu.json 2> $O/bench_insitu.err
python bench.py --steps 996 --warmup 96 --no-cpu-baseline --graph > $O/bench_graph_c3.json 2>/dev/null
python bench.py --workload c5 --steps 600 --warmup 60 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null
python bench.py --workload c5 --steps 600 --warmup 60 --no-cpu-baseline --graph > $O/bench_graph_c5.json 2>/dev/null
python bench.py --workload c2 --frames 4096 --steps 600 --warmup 60 --no-cpu-baseline > $O/bench_c2.json 2>/dev/null
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-delivery > $O/bench_nodelivery.json 2>/dev/null
python tools/host_input_rate.py > $O/host_input_rate.txt 2>&1
timeout -k 10 300 python tools/strain_e2e.py > $O/strain_e2e.json 2>/dev/null
SDR_TAP=256 SDR_TRACE_QUIET=1 SDR_FFT_FPW=1 tools/bin/ft_clock1 2048 > $O/fft_workgroup_spans.txt 2>&1
SDR_FFT_FPW=1 tools/pmc_fft.sh > /dev/null 2>&1 && cp gpurun_out/pmc_fft.txt $O/fft_sq_counters.txt
# what DESIGN.md section 5 quotes about sharing a CU and about the FFT inside the pipeline
tools/bin/ubench_share > $O/ubench_share.txt 2>&1 || true
if [ -f tools/abl/libfftclk.so ]; then SDR_HIP_LIB=$PWD/tools/abl/libfftclk.so python tools/insitu_fft.py > $O/fft_insitu_spans.txt 2>&1 || true; fi
if [ -f tools/abl/libdiag.so ]; then tools/ab_skip.sh 0 254 128 32 2 64 160 0 > $O/skip_matrix.txt 2>&1 || true; fi
echo "all done"
tail -c 600 $O/bench_full.json
