#!/bin/bash
# where a BundleAdjustRecent window call spends its 3.4 ms: Prepare phases, kernels per iteration
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
cat > /tmp/win.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ['R'])
from mcptam_amd import synth, chain_bundle
p = synth.make_config("metric"); w = synth.recent_window(p)
for r in range(6):
    b = chain_bundle.ChainBundle(w.cams, True, True, False)
    ids = w.populate(b); b.Prepare(); rc = b.Compute(10)
    b.GetPoses(ids["mkf"]); b.GetPoints(ids["point"]); b.GetOutlierMeasurements(); b.close()
PY
cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/win_prof
R=$R timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/win_prof -- python /tmp/win.py > $R/gpurun_out/win_prof.log 2>&1; tail -3 $R/gpurun_out/win_prof.log
f=$(find $R/gpurun_out/win_prof -name '*kernel_stats.csv' | head -1); python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(int(r["Calls"]) for r in rows)
print("kernel launches per call:", tot/6.0)
for r in rows[:40]: print("  %-50s calls/call %6.1f avg %7.1f us" % (r["Name"][:50], int(r["Calls"])/6.0, float(r["AverageNs"])/1e3))
PY
find $R/gpurun_out/win_prof -name '*kernel_trace.csv' -delete
