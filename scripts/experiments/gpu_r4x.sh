#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
MCP_HIP_LIB=$R/variants/lib_prrprof.so timeout -k 5 40 python -c "
import sys; sys.path.insert(0,'scripts')
import bench_tracker
bench_tracker.main(frames=2, cpu_frames=0)
" 2>&1 | grep "prr prof" | tail -10
