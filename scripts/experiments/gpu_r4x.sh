#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout -k 5 25 python -m pytest tests/test_img_gpu.py -q -m gpu -x -k "refine or pose_update or track_frame_in_one" 2>&1 | tail -2
bash scripts/gpu_trk_prof.sh 2>&1 | sed -n 3p | cut -d, -f1-4 | cut -c1-20,140-200
