#!/bin/bash
# copy the judged summaries from gpurun_out/ (scratch) into profiles/$ROUND/ (tracked)
set -e
ROUND=${ROUND:-r06}
cd "$(dirname "$0")/.."
mkdir -p profiles/$ROUND
cp gpurun_out/traffic.json profiles/$ROUND/pmc_traffic_per_launch.json
cp gpurun_out/final_bench.json profiles/$ROUND/final_bench.json
f=$(ls -t gpurun_out/prof_final/*/*kernel_stats.csv | head -1); cp "$f" profiles/$ROUND/final_bench_kernel_stats.csv
cp gpurun_out/final_tests.log profiles/$ROUND/final_gpu_tests.log
ls -la profiles/$ROUND
[ -f gpurun_out/trk_kernel_stats.csv ] && cp gpurun_out/trk_kernel_stats.csv profiles/$ROUND/tracker_c3_kernel_stats.csv
[ -f gpurun_out/trk5_kernel_stats.csv ] && cp gpurun_out/trk5_kernel_stats.csv profiles/$ROUND/tracker_c5_kernel_stats.csv
[ -f gpurun_out/trk_trace.txt ] && cp gpurun_out/trk_trace.txt profiles/$ROUND/tracker_c3_frame_timeline.txt
[ -f gpurun_out/window_kernel_stats.csv ] && cp gpurun_out/window_kernel_stats.csv profiles/$ROUND/window_kernel_stats.csv
[ -f gpurun_out/c4_kernel_stats.csv ] && cp gpurun_out/c4_kernel_stats.csv profiles/$ROUND/c4_kernel_stats.csv
[ -s gpurun_out/band_bench.json ] && cp gpurun_out/band_bench.json profiles/$ROUND/band_bench.json
[ -f gpurun_out/window.json ] && cp gpurun_out/window.json profiles/$ROUND/window_bench.json
true
