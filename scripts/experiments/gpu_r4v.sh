#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 300 python scripts/gpu_chol2.py > gpurun_out/chol2.log 2>&1; grep -v "^factor" gpurun_out/chol2.log | tail -22
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()})
"; }
for i in 1 2 3; do timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "auto"; done
