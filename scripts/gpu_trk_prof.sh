#!/bin/bash
# kernel stats of the tracker bench (csv)
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/trk_prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trk_prof -- python $R/scripts/bench_tracker.py > $R/gpurun_out/trk_prof.json 2> $R/gpurun_out/trk_prof.err
echo "rc=$?"
f=$(find $R/gpurun_out/trk_prof -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cut -c1-170 "$f" | head -14; cp "$f" $R/gpurun_out/trk_kernel_stats.csv; fi
find $R/gpurun_out/trk_prof -name '*kernel_trace.csv' -delete
