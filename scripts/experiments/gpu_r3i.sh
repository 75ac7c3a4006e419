#!/bin/bash
# round 3: where does the chunk-major W layout lose?  phase stamps + PMC traffic of both builds
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
cp mcptam_amd/libmcptam_hip.so /tmp/lib_new.so
pmc() {  # $1 = tag
  cd /tmp; export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_$c
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 6 --warmup 1 --cpu-iters 0 --no-roofline > $R/gpurun_out/pmc_$c.log 2>&1
  done
  python $R/scripts/parse_traffic.py $R/gpurun_out > $R/gpurun_out/traffic_$1.json
  python - <<PY
import json
d=json.load(open("$R/gpurun_out/traffic_$1.json"))
for k in ("mcp::k_linearize_group","mcp::k_schur_group","mcp::k_backsub","mcp::k_assemble","void mcp::k_eval<true>"):
    v=d.get(k)
    if v: print("$1", k, "fetch MB %.1f write MB %.1f  (2F+W) MB %.1f  launches %d" % (v["FETCH_SIZE"]/1024, v["WRITE_SIZE"]/1024, (2*v["FETCH_SIZE"]+v["WRITE_SIZE"])/1024, v["launches_FETCH_SIZE"]))
PY
  cd $R
}
pmc new
cp variants/lib_prevW.so mcptam_amd/libmcptam_hip.so
pmc old
cp /tmp/lib_new.so mcptam_amd/libmcptam_hip.so
echo "== stamps, new layout"; bash scripts/sch_prof.sh 2>&1 | tail -6
