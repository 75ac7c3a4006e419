#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_img_gpu.py -m gpu -q -x --timeout 300 --timeout-method=thread -k "many_workgroups" 2>&1 | tail -30
