#!/bin/bash
# dev loop of the one-launch factorisation on the GPU box: correctness against numpy (scripts/gpu_chol2.py), the metric shape's timing,
# the phase stamps of a -DMCP_CP_PROF=2 build (variants/lib_cpprof2.so, scripts/build_variants.sh cpprof2), optionally the solver tests
# usage: chol_dev.sh [tests]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 300 python scripts/gpu_chol2.py 2>&1 | tail -24
for v in cpprof2 cpprof4 cpprof5; do
  [ -f variants/lib_$v.so ] && { echo "== $v"; MCP_HIP_LIB=variants/lib_$v.so timeout 120 python scripts/chol_stamps.py 2>&1 | grep -E "mean over|factor|step 10,|timed out|rror|asked|arrival|wave 2" ; }
done
if [ "$1" = tests ]; then
  timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 --timeout-method=thread -k "cholesky or chol or factorisation or reduced_system or compute_matches or metric_noisy or scheduling_knobs" 2>&1 | tail -5
fi
