#!/bin/bash
# round 2, second GPU pass: GPU suite (strict tracker bytes, destination-ordered staging), kernel-level profile of the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 400 --timeout-method=thread 2>&1 | tail -40 > gpurun_out/r02b_tests.log; tail -12 gpurun_out/r02b_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02b_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02b_prof -- python $R/bench.py --steps 20 --warmup 3 --cpu-iters 0 --no-roofline > $R/gpurun_out/r02b_prof.log 2>&1
echo "stats rc=$?"; tail -1 $R/gpurun_out/r02b_prof.log | cut -c1-300
f=$(ls $R/gpurun_out/r02b_prof/*/*kernel_stats.csv | head -1); head -25 "$f"
cd $R
timeout 300 python bench.py --cpu-iters 0 > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r02b_bench.json
