#!/bin/bash
# round 5, call C: BA suite (new Schur kernel for every batch width, claimed helper entries, wall-clock deadlines), A/B, stamps, workers
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()}, 'fallbacks', d.get('config',{}).get('persist_fallbacks'))
"; }
timeout 700 python -m pytest tests/test_ba_gpu.py -m gpu -q --timeout 300 2>&1 | tail -25 > gpurun_out/r5c_tests.log; cat gpurun_out/r5c_tests.log
for rep in 1 2; do
  for x in 1 0; do MCP_BA_SCHUR4=$x timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "SCHUR4=$x"; done
done
MCP_HIP_LIB=$R/variants/lib_schprof.so timeout 100 python scripts/gpu_quick.py metric 2>&1 | grep "prof\]" | sort | uniq -c | sort -rn | head -3
for w in 111 96 64 32 8; do MCP_BA_CHOL_WORKERS=$w timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "WORKERS=$w"; done
MCP_BA_CHOL_CAPACITY=24 timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "CAPACITY=24"
timeout 120 python scripts/bench_window.py --calls 40 2>/dev/null | cut -c1-420
