#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ba_gpu.py -m gpu -x -q --timeout 400 -k "prepare or parity or golden or calib or identical" 2>&1 | tail -3
bash scripts/variants.sh prevW
bash scripts/variants.sh prevW
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 6 --warmup 1 --cpu-iters 0 --no-roofline > $R/gpurun_out/pmc_$c.log 2>&1
done
python $R/scripts/parse_traffic.py $R/gpurun_out > $R/gpurun_out/traffic_new.json
python - <<PY
import json
d=json.load(open("$R/gpurun_out/traffic_new.json"))
for k in ("mcp::k_linearize_group","mcp::k_schur_group","mcp::k_backsub"):
    v=d.get(k)
    if v: print(k, "fetch MB %.1f write MB %.1f  (2F+W) MB %.1f  launches %d" % (v["FETCH_SIZE"]/1024, v["WRITE_SIZE"]/1024, (2*v["FETCH_SIZE"]+v["WRITE_SIZE"])/1024, v["launches_FETCH_SIZE"]))
PY
