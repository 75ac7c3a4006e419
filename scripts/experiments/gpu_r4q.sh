#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ba_gpu.py -x -q -m gpu > gpurun_out/ba_tests.log 2>&1; tail -8 gpurun_out/ba_tests.log
timeout 300 python bench.py --cpu-iters 0 --no-tracker > gpurun_out/bench_quick.json 2>gpurun_out/bench_quick.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_quick.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline'], d['stages']['ms_total'])"
