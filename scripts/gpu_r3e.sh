#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -q -x --timeout 600 --timeout-method=thread -k "calib or golden or compute_matches or reduced_system or normal_equations or many_poses or long_chains" 2>&1 | tail -8
