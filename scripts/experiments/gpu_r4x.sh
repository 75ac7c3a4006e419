#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for v in 1 0 1 0; do
MCP_BA_HEAD_FINAL=$v timeout -k 5 120 python scripts/bench_window.py --calls 60 > gpurun_out/window.json 2>gpurun_out/window.err; python - <<PY
import json
d=json.load(open('gpurun_out/window.json')); print($v, d['ms_median']['compute_ms'], d['ms_median']['call_ms']); 
PY
done
timeout -k 5 300 python -m pytest tests/test_ba_gpu.py -q -m gpu -k "small_bundle or non_robust or rejected" 2>&1 | tail -3
