#!/bin/bash
# Round-end measurement pass on the MI355X box: full GPU test suite, PMC traffic, rocprofv3 kernel stats, the bench line.
# Everything lands in gpurun_out/ (scratch); scripts/collect_profiles.sh copies the summaries into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 300 --timeout-method=thread 2>&1 | tail -4 > gpurun_out/final_tests.log; cat gpurun_out/final_tests.log
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 6 --warmup 1 --cpu-iters 0 --no-roofline > $R/gpurun_out/pmc_$c.log 2>&1
  echo "$c rc=$?"
done
python $R/scripts/parse_traffic.py $R/gpurun_out > $R/gpurun_out/traffic.json
mkdir -p $R/profiles/r01 && cp $R/gpurun_out/traffic.json $R/profiles/r01/pmc_traffic_per_launch.json      # bench.py reads it from there
rm -rf $R/gpurun_out/prof_final
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --steps 20 --warmup 3 --cpu-iters 0 --no-roofline > $R/gpurun_out/prof_final.log 2>&1
echo "stats rc=$?"; tail -1 $R/gpurun_out/prof_final.log | cut -c1-300
cd $R
timeout 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; cut -c1-700 gpurun_out/final_bench.json
timeout 300 python scripts/bench_tracker.py > gpurun_out/tracker_bench.json 2> gpurun_out/tracker_bench.err; echo "tracker rc=$?"; cut -c1-400 gpurun_out/tracker_bench.json
