#!/bin/bash
# round 5, call D: whole GPU suite, Schur kernel after the flush / prologue / launch-order changes, claim-ahead in the factorisation
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()}, 'fallbacks', d.get('config',{}).get('persist_fallbacks'), 'setup', round(d.get('value_including_setup',{}).get('setup_ms',0),2))
"; }
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -25 > gpurun_out/r5d_tests.log; cat gpurun_out/r5d_tests.log
for rep in 1 2; do
  for x in 1 0; do MCP_BA_SCHUR4_ORDER=$x timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "ORDER=$x"; done
done
MCP_HIP_LIB=$R/variants/lib_schprof.so timeout 100 python scripts/gpu_quick.py metric 2>&1 | grep "prof\]" | sort | uniq -c | sort -rn | head -3
timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['stages']['per_stage']['schur'])); print(json.dumps(d['roofline']))"
