R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 120 python scripts/bench_tracker.py > gpurun_out/tracker_bench.json 2> gpurun_out/tracker_bench.err; cat gpurun_out/tracker_bench.json; tail -3 gpurun_out/tracker_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tr -- python $R/scripts/bench_tracker.py > /dev/null 2>&1
head -14 $R/gpurun_out/prof_tr/*/*kernel_stats.csv | cut -c1-150
