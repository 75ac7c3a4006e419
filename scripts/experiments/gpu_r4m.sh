#!/bin/bash
# round 4: fresh product build -- correctness on all sizes, bench in both stream layouts, BA GPU tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 300 python scripts/gpu_chol2.py > gpurun_out/chol2.log 2>&1; grep -v "^factor\|^solve" gpurun_out/chol2.log | tail -8
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()}, d['config'].get('reduced_system_solves'), d['config'].get('trials_per_iteration'))
"; }
for ov in 0 1 0 1; do MCP_BA_OVERLAP=$ov timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "overlap=$ov persist=1"; done
timeout 900 python -m pytest tests/test_ba_gpu.py -x -q -m gpu > gpurun_out/ba_tests.log 2>&1; tail -5 gpurun_out/ba_tests.log
