#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_img_gpu.py -m gpu -q -x --timeout 600 --timeout-method=thread 2>&1 | tail -30 > gpurun_out/r3d_tests.log; tail -30 gpurun_out/r3d_tests.log
