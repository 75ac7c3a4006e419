#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
MCP_BA_EVT=1 timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline --steps 12 --warmup 2 2> gpurun_out/evt_auto.log >/dev/null
grep "^\[evt\]" gpurun_out/evt_auto.log | tail -12 | cut -c1-300
