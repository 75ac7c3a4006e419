#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s')
"; }
for i in 1 2; do
  timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>/dev/null | show "auto"
  MCP_BA_CHOL_WORKERS=126 timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>/dev/null | show "auto workers=126"
  MCP_BA_OVERLAP=1 MCP_BA_MAIN_SYS=2 timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>/dev/null | show "overlap=1 main_sys=2"
  MCP_BA_OVERLAP=1 MCP_BA_MAIN_SYS=2 MCP_BA_SPEC_DELAY=1 timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>/dev/null | show "overlap=1 main_sys=2 delay=1"
  MCP_BA_OVERLAP=1 MCP_BA_MAIN_SYS=3 timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>/dev/null | show "overlap=1 main_sys=3"
done
MCP_BA_OVERLAP=1 MCP_BA_MAIN_SYS=2 MCP_BA_EVT=1 timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline --steps 8 --warmup 2 2> gpurun_out/evt_d.log >/dev/null
grep "^\[evt\]" gpurun_out/evt_d.log | tail -5 | cut -c1-400
