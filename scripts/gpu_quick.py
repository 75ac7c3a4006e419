"""First-contact script for the GPU box: parity on a small map + stage timings at larger sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from mcptam_amd import synth, chain_bundle
from helpers import run_bundle, compare_runs

print("devices", chain_bundle.device_count(), chain_bundle.last_error())
which = sys.argv[1:] or ["tiny", "c1", "c2", "metric"]
for name in which:
    t0 = time.time()
    p = synth.make_config(name)
    print("== %s: P=%d N=%d M=%d (gen %.1fs)" % (name, p.n_mkf, p.n_points, p.n_meas, time.time() - t0), flush=True)
    iters = 10
    g = chain_bundle.ChainBundle(p.cams, True, True, False, disable_convergence=True, profile=True)
    t0 = time.time(); gpu = run_bundle(g, p, iters); t1 = time.time()
    tm = g.Timing()
    print("gpu rc", gpu["rc"], "wall %.3fs" % (t1 - t0), {k: round(v, 3) if isinstance(v, float) else v for k, v in tm.items()}, flush=True)
    for l in gpu["logs"][:4]:
        print("   ", {k: (float("%.6g" % v) if isinstance(v, float) else v) for k, v in l.items()})
    if name in ("tiny", "c1", "c2"):
        from oracle import OracleBundle
        o = OracleBundle(p.cams, True, True, False); o.DisableConvergence(True)
        t0 = time.time(); ref = run_bundle(o, p, iters); t1 = time.time()
        print("oracle rc", ref["rc"], "wall %.3fs" % (t1 - t0))
        for l in ref["logs"][:4]:
            print("   ", {k: (float("%.6g" % v) if isinstance(v, float) else v) for k, v in l.items()})
        try:
            print("parity", compare_runs(gpu, ref), "outliers equal:", gpu["outliers"] == ref["outliers"], len(gpu["outliers"]))
        except AssertionError as e:
            print("PARITY FAIL", str(e)[:600])
