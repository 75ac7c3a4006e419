#!/bin/bash
# round 4: why is the build without phase stamps slower than the one with them?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
T='import sys, numpy as np; sys.path.insert(0, "scripts"); sys.path.insert(0, "."); import gpu_chol2 as g; from mcptam_amd import chain_bundle as cb
A, b = g.spd(1194, band=6)
for r in range(4):
    tf, tb, x = cb.chol_time(np.tril(A), b, nsys=1, reps=40, band=6); print("factor %.1f us back %.1f us" % (tf*1e3, tb*1e3), flush=True)'
for v in prod cpexp5 cpexp6 cpexp7; do
  echo "== $v"
  if [ $v = prod ]; then timeout 100 python -c "$T" 2>/dev/null; else MCP_HIP_LIB=$R/variants/lib_$v.so timeout 100 python -c "$T" 2>/dev/null; fi
done
