#!/bin/bash
# round 5, call F: structure cache (tests, cold vs cached Prepare), point-order keys without atomics
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -25 > gpurun_out/r5f_tests.log; cat gpurun_out/r5f_tests.log
MCP_BA_TRACE=1 timeout 200 python scripts/setup_time.py 2>&1 | grep -v "streams\]" | head -64
timeout 200 python bench.py --cpu-iters 0 --no-tracker 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], json.dumps(d['value_including_setup'])); print(json.dumps(d['config']['setup_outside_timed_region'])); print(json.dumps(d.get('recent_window'))[:900])"
timeout 120 python scripts/bench_window.py --calls 40 2>/dev/null | cut -c1-420
MCP_BA_STRUCT_CACHE=0 timeout 120 python scripts/bench_window.py --calls 40 2>/dev/null | cut -c1-420
