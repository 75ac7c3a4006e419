#!/bin/bash
# round 4, call 2: phase stamps of the one-launch factorisation, the BA GPU tests on it, bench A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
MCP_HIP_LIB=$R/variants/lib_cpprof.so timeout 200 python scripts/gpu_chol2.py 1194 > gpurun_out/cpprof.log 2>&1
grep -v "^factor\|^solve" gpurun_out/cpprof.log | head -70
for v in 1 0 1 0; do
  MCP_BA_CHOL_PERSIST=$v timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>gpurun_out/bench_p$v.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('persist=$v', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()}, d['config'].get('reduced_system_solves'))
"
done
timeout 900 python -m pytest tests/test_ba_gpu.py -x -q -m gpu > gpurun_out/ba_tests.log 2>&1; tail -15 gpurun_out/ba_tests.log
