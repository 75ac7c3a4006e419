#!/bin/bash
# round 5, call I: batched assembly (bit-identity tests + A/B)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()}, 'fallbacks', d.get('config',{}).get('persist_fallbacks'))
"; }
timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -8 > gpurun_out/r5i_tests.log; cat gpurun_out/r5i_tests.log
for rep in 1 2; do
  for x in 1 0; do MCP_BA_ASM_BATCH=$x timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "ASM_BATCH=$x"; done
done
