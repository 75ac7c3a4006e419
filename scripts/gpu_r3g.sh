#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
run() { env "$@" timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>gpurun_out/r3g.err | python -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
if not t: print('$*', 'NO OUTPUT'); sys.exit()
d=json.loads(t[-1])
print('$*', round(d['value'],1), 'it/s')"; grep "streams\]" gpurun_out/r3g.err | head -14 | tr '\n' ';'; echo; }
for rep in 1 2; do
run MCP_BA_TRACE=1 MCP_BA_SPEC_TRIALS=1
run MCP_BA_TRACE=1 MCP_BA_SPEC_TRIALS=2
run MCP_BA_SPEC_TRIALS=2 MCP_BA_SPECULATE=2
done
