#!/bin/bash
# device timeline of one tracker frame (c3, or `c5` as argument; one submission per frame): kernels + copies in start order with gaps
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/trk_trace
TRK_FUSED_ONLY=1 timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trk_trace -- python $R/scripts/bench_tracker.py $1 > /dev/null 2> /tmp/trk_trace.err
echo "rc=$?"
python $R/scripts/trk_trace_report.py /tmp/trk_trace | tee $R/gpurun_out/trk_trace.txt
