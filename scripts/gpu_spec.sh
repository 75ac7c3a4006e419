#!/bin/bash
# A/B of the speculative second system: BA parity tests, then the bench with and without it
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ba_gpu.py -m gpu -x -q --timeout 300 --timeout-method=thread > gpurun_out/spec_tests.log 2>&1
tail -3 gpurun_out/spec_tests.log
for s in 3 2 1; do
  MCP_BA_SPECULATE=$s timeout 300 python bench.py --cpu-iters 0 > gpurun_out/spec_bench_$s.json 2> gpurun_out/spec_bench_$s.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/spec_bench_$s.json").read().strip().splitlines()[-1])
print("speculate=$s value %.1f ms/step %.3f"%(d["value"], d["ms_per_step"]), {k:round(v,2) for k,v in d.get("stages",{}).get("ms_total",{}).items()})
PY
done
