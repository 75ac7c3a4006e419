#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
echo "== c2small, poison"; MCP_DEV_CACHE_POISON=1 timeout -k 5 200 python scripts/experiments/dbg_cache.py c2small 2>&1 | tail -8 | cut -c1-400
echo "== tests, poison"; MCP_DEV_CACHE_POISON=1 timeout -k 5 500 python -m pytest tests/test_ba_gpu.py -q -m gpu 2>&1 | tail -8
