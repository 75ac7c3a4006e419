#!/bin/bash
# GPU-box job: parity tests, bench line, rocprofv3 kernel trace (run via gpurun from the repo root)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 20 --warmup 3 --cpu-iters 0 --no-roofline > $R/gpurun_out/prof.log 2>&1; echo "rocprof rc=$?"
find $R/gpurun_out/prof -name "*stats*" | head
