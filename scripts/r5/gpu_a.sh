#!/bin/bash
# round 5, call A: BA suite with the new Schur kernel, A/B against the old one, stamps, 1-WG-per-CU variant
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()}, 'parity', d.get('parity_at_metric',{}).get('ok'))
"; }
timeout 700 python -m pytest tests/test_ba_gpu.py -m gpu -x -q --timeout 300 2>&1 | tail -15 > gpurun_out/r5a_tests.log; cat gpurun_out/r5a_tests.log
for rep in 1 2; do
  for x in 1 0; do MCP_BA_SCHUR4=$x timeout 200 python bench.py --cpu-iters 0 2>/dev/null | show "SCHUR4=$x"; done
done
MCP_HIP_LIB=$R/variants/lib_s4w1.so timeout 200 python bench.py --cpu-iters 0 2>/dev/null | show "s4w1"
MCP_BA_POINT_ORDER=0 MCP_BA_GROUP_LMAX13=0 MCP_BA_SCHUR4=0 timeout 200 python bench.py --cpu-iters 0 2>/dev/null | show "r4-structure"
MCP_HIP_LIB=$R/variants/lib_schprof.so timeout 100 python scripts/gpu_quick.py metric 2>&1 | grep "prof\]" | sort | uniq -c | sort -rn | head -8
