#!/bin/bash
# HBM traffic of the bench kernels from the PMC counters (MI355X_MICROARCH.md "HBM": separate --pmc passes,
# FETCH_SIZE/WRITE_SIZE in KiB-units of the TCC EA requests; FETCH_SIZE x2 on gfx950 for wide streaming reads).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 6 --warmup 1 --cpu-iters 0 --no-roofline > $R/gpurun_out/pmc_$c.log 2>&1
  echo "$c rc=$?"; ls $R/gpurun_out/pmc_$c/*/ | head
done
python $R/scripts/parse_traffic.py $R/gpurun_out > $R/gpurun_out/traffic.json; cat $R/gpurun_out/traffic.json | head -40
