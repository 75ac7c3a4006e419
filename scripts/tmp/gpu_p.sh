#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()})
"; }
for rep in 1 2; do
timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "base"
MCP_BA_ASM_LONG=1 timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "ASM_LONG=1"
MCP_BA_CHOL_CAPACITY=600 MCP_BA_CHOL_WORKERS=126 timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "WORKERS=126"
MCP_BA_CHOL_CAPACITY=560 MCP_BA_CHOL_WORKERS=119 timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "WORKERS=119"
done
