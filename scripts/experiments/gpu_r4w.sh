#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout -k 5 120 python scripts/bench_window.py --trace > gpurun_out/window.json 2>gpurun_out/window.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/window.json')); print(d['ms_median']); 
PY
grep "^\[evt\]" gpurun_out/window.err | tail -5
timeout -k 5 400 python -m pytest tests/test_ba_gpu.py -q -m gpu 2>&1 | tail -6
timeout -k 5 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d.get('stages',{}).get('ms_total'))"
