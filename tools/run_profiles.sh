#!/bin/bash
# Runs on the GPU box (from the repo root): everything tools/make_profiles.py condenses into profiles/.
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
set -e
rm -rf gpurun_out/prof_r01 gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01 -o r01 --output-format csv -- python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/rocprof.log 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --serial > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --serial > gpurun_out/pmc_write.log 2>&1
echo "pmc done"
python bench.py > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err
echo "bench done"
python bench.py --steps 300 --warmup 30 --kernel-breakdown --no-cpu-baseline --serial > gpurun_out/bench_serial.log 2> gpurun_out/bench_serial.err
python bench.py --steps 300 --warmup 30 --kernel-breakdown --no-cpu-baseline > gpurun_out/bench_insitu.log 2> gpurun_out/bench_insitu.err
tail -1 gpurun_out/bench_full.log | cut -c1-600
