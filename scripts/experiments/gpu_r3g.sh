#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
run() { env "$@" timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>gpurun_out/r3g.err | python -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
if not t: print('$*', 'NO OUTPUT'); sys.exit()
d=json.loads(t[-1])
print('$*', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d['stages']['ms_total'].items()})"; }
for rep in 1 2; do
run MCP_BA_SPEC_CUS=0
run MCP_BA_SPEC_CUS=128
run MCP_BA_SPEC_CUS=192
run MCP_BA_SPEC_CUS=64
done
