#!/bin/bash
# kernel stats of the bench (rocprofv3), headline workload only
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_mid
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mid -- python $R/bench.py --steps 20 --warmup 3 --cpu-iters 0 --no-roofline > $R/gpurun_out/prof_mid.log 2>&1
echo "stats rc=$?"
f=$(find $R/gpurun_out/prof_mid -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/mid_kernel_stats.csv; find $R/gpurun_out/prof_mid -name '*kernel_trace.csv' -delete
column -s, -t < $R/gpurun_out/mid_kernel_stats.csv | cut -c1-150 | head -30
