#!/bin/bash
# quick GPU check: a subset of the GPU tests (pytest -k expression in $1, default: the solver-chain tests) and the bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
K=${1:-"cholesky or compute_matches or reduced_system or c2_full or metric_noisy"}
timeout 600 python -m pytest tests -m gpu -q --timeout 300 --timeout-method=thread -k "$K" 2>&1 | tail -12 > gpurun_out/quick_tests.log; tail -6 gpurun_out/quick_tests.log
timeout 300 python bench.py --cpu-iters 0 > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/quick_bench.json").read().strip().splitlines()[-1])
print(round(d["value"],1), "it/s", {k: round(v,2) for k,v in d["stages"]["ms_total"].items()}, d["config"].get("reduced_system_solves"), d["config"].get("trials_served_speculatively"))
PY
