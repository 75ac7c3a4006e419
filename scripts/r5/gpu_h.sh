#!/bin/bash
# round 5, call H: PMC traffic of the BA kernels (two separate passes), kernel stats of the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 6 --warmup 1 --cpu-iters 0 --no-roofline --no-tracker > $R/gpurun_out/pmc_$c.log 2>&1
  echo "$c rc=$?"
done
python $R/scripts/parse_traffic.py $R/gpurun_out > $R/gpurun_out/traffic.json
python - <<PY
import json
t=json.load(open("$R/gpurun_out/traffic.json"))
for k,v in sorted(t.items(), key=lambda kv: -(2*kv[1].get("FETCH_SIZE",0)+kv[1].get("WRITE_SIZE",0))):
    print("%-60s fetch %9.0f KB  write %9.0f KB  -> %.1f MB/launch (%d launches)" % (k[:60], v.get("FETCH_SIZE",0), v.get("WRITE_SIZE",0), (2*v.get("FETCH_SIZE",0)+v.get("WRITE_SIZE",0))/1024, v.get("launches_FETCH_SIZE",0)))
PY
rm -rf $R/gpurun_out/prof_final
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --steps 20 --warmup 3 --cpu-iters 0 --no-roofline --no-tracker > $R/gpurun_out/prof_final.log 2>&1
f=$(find $R/gpurun_out/prof_final -name '*kernel_stats.csv' | head -1); cut -c1-150 $f | head -24
find $R/gpurun_out/prof_final -name '*kernel_trace.csv' -delete; find $R/gpurun_out -name '*counter_collection.csv' -size +20M -delete
