#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()})
"; }
MCP_BA_SMALL_POINTS=100000 timeout -k 5 150 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "metric quad groups"
timeout -k 5 150 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "metric full groups"
