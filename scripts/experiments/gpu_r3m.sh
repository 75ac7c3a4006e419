#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
MCP_BA_TRACE=1 timeout 300 python scripts/setup_time.py 2>&1 | grep -B45 "recent window" | grep -A45 "metric rep 1" | head -60
nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"
