#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s')
"; }
for i in 1 2; do timeout -k 5 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "merged"; done
for i in 1 2; do MCP_BA_SMALL=0 timeout -k 5 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | show "separate"; done
timeout -k 5 300 python -m pytest tests/test_ba_gpu.py -q -m gpu -k "scheduling_knobs or small_bundle or oracle" 2>&1 | tail -3
