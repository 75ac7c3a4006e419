#!/bin/bash
# round 5, call E: dephased Schur workgroups, Prepare phases, stamps of the panel with and without its bulk updates
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()}, 'fallbacks', d.get('config',{}).get('persist_fallbacks'), 'setup', round(d.get('value_including_setup',{}).get('setup_ms',0),2))
"; }
for rep in 1 2; do
  timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "base"
  for v in s4d1 s4d2 s4d3; do MCP_HIP_LIB=$R/variants/lib_$v.so timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "$v"; done
done
MCP_HIP_LIB=$R/variants/lib_s4d2p.so timeout 100 python scripts/gpu_quick.py metric 2>&1 | grep "prof\]" | sort | uniq -c | sort -rn | head -3
MCP_BA_TRACE=1 timeout 200 python scripts/setup_time.py 2>&1 | head -50
for v in cpprof4 cpabl1; do echo "== $v"; MCP_HIP_LIB=$R/variants/lib_$v.so timeout 100 python scripts/r5/chol_stamps.py 2>&1 | grep "mean over\|factor"; done
