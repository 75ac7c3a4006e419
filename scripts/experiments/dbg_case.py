import os, sys, hashlib
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from mcptam_amd import synth, chain_bundle
from helpers import run_bundle
case = sys.argv[1]
if case == "metric_step":
    os.environ["MCP_BA_CHOL_PERSIST"] = "0"; p = synth.make_config("metric"); it = 4
elif case == "c2_step":
    os.environ["MCP_BA_CHOL_PERSIST"] = "0"; p = synth.make_config("c2"); it = 4
elif case == "c2small_plain":
    os.environ["MCP_BA_SMALL"] = "0"; p = synth.make_config("c2", n_mkf=12, n_points=1500); it = 4
r = None
for rep in range(2):
    r = run_bundle(chain_bundle.ChainBundle(p.cams, True, True, False, disable_convergence=True), p, it)
print([(l["trials"], l["accepted"]) for l in r["logs"]], hashlib.md5(r["X"].tobytes()).hexdigest()[:8])
