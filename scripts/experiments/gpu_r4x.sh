#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout -k 5 200 python -m pytest tests/test_ba_gpu.py -q -m gpu -k "quarter_groups or small_bundle" 2>&1 | tail -4
