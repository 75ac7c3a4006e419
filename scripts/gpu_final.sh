#!/bin/bash
# Round-end measurement pass on the MI355X box: full GPU test suite, PMC traffic (two separate rocprofv3 --pmc passes, as
# MI355X_MICROARCH.md prescribes), rocprofv3 kernel stats of the bench, the bench line (incl. CPU baselines and the tracker line).
# Everything lands in gpurun_out/ (scratch, merged back); scripts/collect_profiles.sh copies the summaries into profiles/$ROUND/.
ROUND=${ROUND:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -q --timeout 400 --timeout-method=thread 2>&1 | tail -6 > gpurun_out/final_tests.log; cat gpurun_out/final_tests.log
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 6 --warmup 1 --cpu-iters 0 --no-roofline > $R/gpurun_out/pmc_$c.log 2>&1
  echo "$c rc=$?"
done
python $R/scripts/parse_traffic.py $R/gpurun_out > $R/gpurun_out/traffic.json
mkdir -p $R/profiles/$ROUND && cp $R/gpurun_out/traffic.json $R/profiles/$ROUND/pmc_traffic_per_launch.json      # bench.py reads it from there (this box's copy)
rm -rf $R/gpurun_out/prof_final
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --steps 20 --warmup 3 --cpu-iters 0 --no-roofline > $R/gpurun_out/prof_final.log 2>&1
echo "stats rc=$?"; tail -1 $R/gpurun_out/prof_final.log | cut -c1-300
cd $R
timeout -k 10 500 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; cut -c1-900 gpurun_out/final_bench.json; tail -2 gpurun_out/final_bench.err
bash $R/scripts/gpu_trk_prof.sh > gpurun_out/trk_prof_stdout.log 2>&1; tail -3 gpurun_out/trk_prof_stdout.log
# kernel stats of the c5 tracker frame (eight 1280x960 cameras, 8000 points, one device)
cd /tmp; rm -rf $R/gpurun_out/trk5_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trk5_prof -- python $R/scripts/bench_tracker.py c5 > $R/gpurun_out/trk5_prof.json 2> $R/gpurun_out/trk5_prof.err
f=$(find $R/gpurun_out/trk5_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/trk5_kernel_stats.csv
find $R/gpurun_out/trk5_prof -name '*kernel_trace.csv' -delete; cd $R
# kernel stats of the c4 map (500 MKF, 100k points, 800k measurements) as a whole on this one device
cd /tmp; rm -rf $R/gpurun_out/prof_c4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4 -- python $R/scripts/bench_secondary.py c4 > $R/gpurun_out/prof_c4.json 2> $R/gpurun_out/prof_c4.err
f=$(find $R/gpurun_out/prof_c4 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/c4_kernel_stats.csv
find $R/gpurun_out/prof_c4 -name '*kernel_trace.csv' -delete; cd $R
# kernel stats + whole-call times of the BundleAdjustRecent window (the call MCPTAM makes most)
cd /tmp; rm -rf $R/gpurun_out/prof_window
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_window -- python $R/scripts/bench_window.py --calls 10 > $R/gpurun_out/prof_window.log 2>&1
f=$(find $R/gpurun_out/prof_window -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/window_kernel_stats.csv
find $R/gpurun_out/prof_window -name '*kernel_trace.csv' -delete; cd $R
timeout -k 10 120 python scripts/bench_window.py --calls 40 > gpurun_out/window.json 2> gpurun_out/window.err; cut -c1-400 gpurun_out/window.json
# a band trajectory of the headline's sizes: the factorisation as two chains and as one
( timeout -k 10 200 python scripts/bench_band.py arc; timeout -k 10 200 python scripts/bench_band.py ring; timeout -k 10 200 python scripts/bench_band.py metric_shuffled ) > gpurun_out/band_bench.json 2> gpurun_out/band_bench.err; cut -c1-300 gpurun_out/band_bench.json
# device timeline of one c3 tracker frame (kernels + copies in start order with gaps)
bash $R/scripts/gpu_trk_trace.sh > gpurun_out/trk_trace_stdout.log 2>&1; tail -9 gpurun_out/trk_trace.txt
