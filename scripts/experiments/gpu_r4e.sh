#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_img_gpu.py -m gpu -x -q --timeout 400 2>&1 | tail -2
timeout 600 python -m pytest tests/test_ba_gpu.py -m gpu -x -q --timeout 400 -k "golden or select or median or sigma or parity" 2>&1 | tail -1
MCP_TRACK_REFINE_GATHER=1 timeout 300 python -m pytest tests/test_img_gpu.py -m gpu -x -q --timeout 400 -k "many_workgroups" 2>&1 | tail -1
bash scripts/experiments/gpu_r3t.sh
