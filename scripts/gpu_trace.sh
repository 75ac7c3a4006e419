#!/bin/bash
# kernel trace of the bench (csv) for offline critical-path analysis (scripts/trace_report.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -- python $R/bench.py --steps 20 --warmup 3 --cpu-iters 0 --no-roofline --no-tracker > $R/gpurun_out/trace.log 2>&1
echo "trace rc=$?"
f=$(find $R/gpurun_out/trace -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python $R/scripts/trace_report.py "$f" | tee $R/gpurun_out/trace_report.txt
