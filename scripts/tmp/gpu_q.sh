#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
MCP_DEV_CACHE_POISON=1 timeout 1200 python -m pytest tests/test_ba_gpu.py -m gpu -q --timeout 400 2>&1 | tail -15
