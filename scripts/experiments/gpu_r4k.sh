#!/bin/bash
# round 4, call 5
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 300 python scripts/gpu_chol2.py 1 33 65 97 130 500 1194 > gpurun_out/chol2.log 2>&1; grep -v "^factor\|^solve" gpurun_out/chol2.log | tail -8
MCP_HIP_LIB=$R/variants/lib_cpprof.so timeout 200 python scripts/gpu_chol2.py 1194 > gpurun_out/cpprof.log 2>&1
grep -A20 "cp prof" gpurun_out/cpprof.log | sed -n 22,44p
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()}, d['config'].get('reduced_system_solves'), d['config'].get('trials_per_iteration'))
"; }
for ov in 0 1 0 1; do MCP_BA_OVERLAP=$ov timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "overlap=$ov persist=1"; done
