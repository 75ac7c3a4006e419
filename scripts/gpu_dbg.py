import os, sys, time, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcptam_amd import synth, chain_bundle
from oracle import OracleBundle
name = sys.argv[1]
p = synth.make_config(name)
print("cfg", name, p.n_meas, flush=True)
g = chain_bundle.ChainBundle(p.cams, True, True, False)
p.populate(g)
print("populated", flush=True)
n = g.Prepare(); print("prepared", n, flush=True)
c, e = g.Eval(p.n_meas); print("eval ok", np.abs(c).max(), flush=True)
print("robust", g.DebugRobustChi2(), flush=True)
x = g.DebugSolve(1e-3); print("solve ok", np.abs(x).max(), flush=True)
o = OracleBundle(p.cams, True, True, False); p.populate(o)
rc, xs, xd = o.DebugSolve(1e-3)
print("solve parity", np.abs(x - xs).max() / np.abs(xs).max(), flush=True)
t = time.time(); rc = g.Compute(5); print("compute", rc, time.time() - t, [l["trials"] for l in g.IterLogs()], flush=True)
