#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cat > /tmp/prep.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ['R'])
import numpy as np
from mcptam_amd import synth, chain_bundle
p = synth.make_config("metric")
for lib in sys.argv[1:]:
    pt, at = [], []
    for r in range(12):
        b = chain_bundle.ChainBundle(p.cams, True, True, False, disable_convergence=True)
        p.populate(b); t1 = time.perf_counter(); b.Prepare(); t2 = time.perf_counter()
        pt.append((t2 - t1)*1e3); at.append(b.abi_seconds*1e3)
        b.Compute(2); b.close()
    print(lib, "prepare ms:", " ".join("%.1f" % v for v in pt), "| median %.2f" % np.median(pt[2:]), "| replay median %.2f" % np.median(at[2:]), flush=True)
PY
R=$R python /tmp/prep.py new
cp mcptam_amd/libmcptam_hip.so /tmp/new.so; cp variants/lib_prevW.so mcptam_amd/libmcptam_hip.so
R=$R python /tmp/prep.py prev
cp /tmp/new.so mcptam_amd/libmcptam_hip.so
R=$R MCP_BA_TRACE=1 python /tmp/prep.py new 2>&1 | grep -E "by point|slot threads|alloc\+upload|state|groups|local indices|sort\+slots" | tail -21
