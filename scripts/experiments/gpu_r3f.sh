#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_img_gpu.py -m gpu -q -x --timeout 300 --timeout-method=thread -k "pose_refine or c5 or camera_per_rank" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for ppw in 256 512 1024; do
rm -rf $R/gpurun_out/prof_c5
MCP_TRACK_REFINE_PPW=$ppw timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -- python $R/scripts/bench_tracker.py c5 > $R/gpurun_out/prof_c5.log 2>&1
f=$(ls -t $R/gpurun_out/prof_c5/*/*kernel_stats.csv | head -1); echo "ppw=$ppw $(grep k_pose_refine_multi $f | sed 's/.*)",//')"
done
