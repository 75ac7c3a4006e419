#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()}, d['config'].get('reduced_system_solves'), d['config'].get('trials_per_iteration'))
"; }
for i in 1 2 3; do timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "auto"; MCP_BA_OVERLAP=0 timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "overlap=0"; done
MCP_BA_EVT=1 timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline --steps 8 --warmup 2 2> gpurun_out/evt_auto.log >/dev/null
grep "^\[evt\]" gpurun_out/evt_auto.log | tail -7 | cut -c1-700
