#!/bin/bash
# A/B of an environment switch on the bench: usage gpu_ab.sh VAR val1 val2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
V=$1; shift
for x in "$@"; do
  for rep in 1 2; do
  env $V=$x timeout 200 python bench.py --cpu-iters 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V=$x', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d['stages']['ms_total'].items()})"
  done
done
