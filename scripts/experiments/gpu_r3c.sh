#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_cpp_host.py -m gpu -q -x --timeout 600 --timeout-method=thread 2>&1 | tail -5
MCP_BA_TRACE=1 timeout 300 python scripts/setup_time.py 2>&1 | tee gpurun_out/r3c_setup.log | grep -v "^\[mcp_ba prepare\]" | head -8
timeout 300 python bench.py --cpu-iters 0 > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3c_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c_bench.json").read().strip().splitlines()[-1])
print(round(d["value"],1), "it/s", d.get("value_including_setup"), d["config"]["setup_outside_timed_region"])
print(d.get("recent_window"))
PY
