#!/bin/bash
# round 5, call G: whole GPU suite (concurrency test, deadline in the tracker's grid barrier), cold Prepare phases in steady state, populate split
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -25 > gpurun_out/r5g_tests.log; cat gpurun_out/r5g_tests.log
SETUP_COLD=1 MCP_BA_TRACE=1 timeout 200 python scripts/setup_time.py 2>&1 | grep -v "streams\]" | sed -n 20,62p
