#!/bin/bash
# linearize() waits only for the readers of the previous linearisation: BA suite, A/B (MCP_BA_LIN_JOIN=1 = the full join), timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -x -q --timeout 400 2>&1 | tail -2
bash scripts/gpu_ab.sh MCP_BA_LIN_JOIN 1 0 1 0
MCP_BA_EVT=1 timeout 300 python bench.py --steps 8 --warmup 4 --cpu-iters 0 --no-roofline 2>&1 | grep "\[evt\] iter" | tail -4 | cut -c1-200
