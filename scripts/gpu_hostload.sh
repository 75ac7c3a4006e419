#!/bin/bash
# the bench next to a loaded host (N busy processes): how much of the LM rate depends on the host keeping up, with and without
# graph replay of the factorisation chains
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
N=${1:-16}
pids=""
for i in $(seq $N); do python -c "
while True: pass" & pids="$pids $!"; done
sleep 1
for g in 0 1 0 1; do
  MCP_BA_GRAPH=$g timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('loaded host ($N busy), graph=$g', round(d['value'],1), 'it/s')"
done
kill $pids 2>/dev/null
wait 2>/dev/null
for g in 0 1; do
  MCP_BA_GRAPH=$g timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('quiet host, graph=$g', round(d['value'],1), 'it/s')"
done
