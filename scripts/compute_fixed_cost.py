"""Fixed cost of a Compute(K) call at the metric size: T(K) = a + b K, from calls of different K on one prepared handle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcptam_amd import chain_bundle, synth
p = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "metric")
b = chain_bundle.ChainBundle(p.cams, True, True, False, disable_convergence=True)
p.populate(b); b.Prepare()
b.Compute(40)
rows = []
for rep in range(3):
    for k in (1, 2, 5, 10, 20, 40):
        t0 = time.perf_counter(); rc = b.Compute(k); dt = (time.perf_counter() - t0)*1e3
        rows.append((k, dt))
        print("Compute(%2d) = %d: %.3f ms  (%.3f ms per iteration)" % (k, rc, dt, dt/k), flush=True)
K = np.array([r[0] for r in rows], float); T = np.array([r[1] for r in rows])
A = np.stack([np.ones_like(K), K], 1); a, bb = np.linalg.lstsq(A, T, rcond=None)[0]
print("fit: T(K) = %.3f ms + %.4f ms K" % (a, bb))
if os.environ.get("MCP_BA_TRACE_COMPUTE"):
    b.Compute(3)
b.close()
