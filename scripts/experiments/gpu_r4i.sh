#!/bin/bash
# round 4, call 3: is the one-launch factorisation's gain eaten by the speculative launch's footprint?  (A) no speculation, (B) event timelines
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()}, d['config'].get('reduced_system_solves'), d['config'].get('trials_per_iteration'))
"; }
for v in 1 0 1 0; do MCP_BA_SPECULATE=0 MCP_BA_CHOL_PERSIST=$v timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "speculate=0 persist=$v"; done
for v in 1 0; do MCP_BA_OVERLAP=0 MCP_BA_CHOL_PERSIST=$v timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "overlap=0 persist=$v"; done
for v in 1 0; do
  MCP_BA_EVT=1 MCP_BA_CHOL_PERSIST=$v timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline --steps 8 --warmup 2 2> gpurun_out/evt_p$v.log >/dev/null
  echo "== timeline persist=$v"; grep "^\[evt\]" gpurun_out/evt_p$v.log | tail -7 | cut -c1-900
done
