#!/bin/bash
# copy the judged summaries from gpurun_out/ (scratch) into profiles/r01/ (tracked)
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/r01
cp gpurun_out/traffic.json profiles/r01/pmc_traffic_per_launch.json
cp gpurun_out/final_bench.json profiles/r01/final_bench.json
f=$(ls -t gpurun_out/prof_final/*/*kernel_stats.csv | head -1); cp "$f" profiles/r01/final_bench_kernel_stats.csv
[ -s gpurun_out/tracker_bench.json ] && cp gpurun_out/tracker_bench.json profiles/r01/tracker_c3_bench.json
cp gpurun_out/final_tests.log profiles/r01/final_gpu_tests.log
ls -la profiles/r01
