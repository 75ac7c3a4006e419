#!/bin/bash
# round 4, call 1: bring-up of the one-launch factorisation
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
( timeout 300 python scripts/gpu_chol2.py; echo "--- per-step kernels"; MCP_BA_CHOL_PERSIST=0 timeout 200 python scripts/gpu_chol2.py 1194 ) > gpurun_out/chol2.log 2>&1
tail -60 gpurun_out/chol2.log
