#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -q -x --timeout 600 --timeout-method=thread -k "failed_factorisation or abort_flag or scheduling_knobs" 2>&1 | tail -12
