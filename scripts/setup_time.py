"""What a BundleAdjust() call pays before its first LM iteration: the C-ABI replay of the map (populate) and Prepare()
(structure build + upload), for the metric map and for the BundleAdjustRecent window of it.  MCP_BA_TRACE=1 prints Prepare's phases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcptam_amd import synth, chain_bundle

def run(name, p, reps=3):
    for r in range(reps):
        if os.environ.get("SETUP_COLD"):
            chain_bundle.struct_cache_clear()          # every repetition on an empty structure cache (a topology the process has not seen)
        b = chain_bundle.ChainBundle(p.cams, True, True, False, disable_convergence=True)
        t0 = time.perf_counter(); p.populate(b); t1 = time.perf_counter()
        b.Prepare(); t2 = time.perf_counter()
        rc = b.Compute(10); t3 = time.perf_counter()
        print("%s rep %d: populate %.2f ms (inside the library %.2f ms), prepare %.2f ms, 10 iterations %.2f ms (rc %d)" % (
            name, r, (t1 - t0)*1e3, b.abi_seconds*1e3, (t2 - t1)*1e3, (t3 - t2)*1e3, rc), flush=True)
        b.close()

p = synth.make_config("metric")
run("metric", p)
w = synth.recent_window(p)
print("recent window: %d MKF (%d free), %d points, %d measurements" % (w.n_mkf, int((~w.base_fixed).sum()), w.n_points, w.n_meas))
run("recent", w)
