#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout -k 5 120 python scripts/bench_window.py > gpurun_out/window.json 2>gpurun_out/window.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/window.json')); print(d['ms_median']); 
PY
timeout -k 5 300 python -m pytest tests/test_ba_gpu.py -q -m gpu -k "split_assembly or small_bundle or oracle or debug or system" 2>&1 | tail -4
