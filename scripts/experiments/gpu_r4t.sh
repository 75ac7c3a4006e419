#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 300 python scripts/gpu_chol2.py > gpurun_out/chol2.log 2>&1; grep -v "^factor\|^solve" gpurun_out/chol2.log | tail -6
T='import sys, numpy as np; sys.path.insert(0, "scripts"); sys.path.insert(0, "."); import gpu_chol2 as g; from mcptam_amd import chain_bundle as cb
A, b = g.spd(1194, band=6)
for nsys in (1, 3):
    tf, tb, x = cb.chol_time(np.tril(A), b, nsys=nsys, reps=40, band=6); print("nsys %d factor %.1f us back %.1f us" % (nsys, tf*1e3, tb*1e3), flush=True)'
for w in 24 32 48 64 96 126; do echo "== workers $w"; MCP_BA_CHOL_WORKERS=$w timeout 100 python -c "$T" 2>&1 | tail -2; done
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()}, d['config'].get('reduced_system_solves'), d['config'].get('trials_per_iteration'))
"; }
for i in 1 2; do
  timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>/dev/null | show "auto"
  MCP_BA_OVERLAP=1 MCP_BA_SPEC_DELAY=1 timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>/dev/null | show "overlap=1 delay=1"
  MCP_BA_OVERLAP=1 timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>/dev/null | show "overlap=1"
  MCP_BA_CHOL_WORKERS=32 MCP_BA_OVERLAP=1 MCP_BA_SPEC_DELAY=1 timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>/dev/null | show "workers=32 overlap=1 delay=1"
  MCP_BA_CHOL_WORKERS=32 MCP_BA_OVERLAP=1 timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline 2>/dev/null | show "workers=32 overlap=1"
done
MCP_BA_OVERLAP=1 MCP_BA_SPEC_DELAY=1 MCP_BA_EVT=1 timeout 200 python bench.py --cpu-iters 0 --no-tracker --no-roofline --steps 8 --warmup 2 2> gpurun_out/evt_d.log >/dev/null
grep "^\[evt\]" gpurun_out/evt_d.log | tail -5 | cut -c1-400
