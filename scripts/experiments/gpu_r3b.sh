#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -q -x --timeout 600 --timeout-method=thread 2>&1 | tail -15 > gpurun_out/r3b_tests.log; tail -6 gpurun_out/r3b_tests.log
MCP_BA_TRACE=1 timeout 300 python scripts/setup_time.py 2>&1 | tee gpurun_out/r3b_setup.log | grep -v "^\[mcp_ba prepare\]" | tail; grep "prepare\]" gpurun_out/r3b_setup.log | sed -n 14,28p
