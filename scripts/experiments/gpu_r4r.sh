#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ba_gpu.py tests/test_img_gpu.py -x -q -m gpu -k "partitioned or spawns or camera_per_rank_frame or failed_factorisation" > gpurun_out/new_tests.log 2>&1; tail -30 gpurun_out/new_tests.log
