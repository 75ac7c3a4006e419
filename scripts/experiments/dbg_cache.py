import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from mcptam_amd import synth, chain_bundle
from helpers import run_bundle

def gpu(cams): return chain_bundle.ChainBundle(cams, True, True, False, disable_convergence=True)
def br(r): return [(l["trials"], l["accepted"]) for l in r["logs"]]
which = sys.argv[1]
if which == "metric":
    p = synth.make_config("metric")
    for rep in range(2):
        os.environ.pop("MCP_BA_CHOL_PERSIST", None)
        a = run_bundle(gpu(p.cams), p, 6)
        os.environ["MCP_BA_CHOL_PERSIST"] = "0"
        b = run_bundle(gpu(p.cams), p, 6)
        print("persist", br(a)); print("step   ", br(b)); print("max dX", np.abs(a["X"] - b["X"]).max(), "chi", [l["chi2_end"] for l in a["logs"]][-1], [l["chi2_end"] for l in b["logs"]][-1])
else:
    p = synth.make_config("c2", n_mkf=12, n_points=1500)
    for rep in range(2):
        os.environ.pop("MCP_BA_SMALL", None)
        a = run_bundle(gpu(p.cams), p, 10)
        os.environ["MCP_BA_SMALL"] = "0"
        b = run_bundle(gpu(p.cams), p, 10)
        print("small", br(a)); print("plain", br(b))
        for i, (x, y) in enumerate(zip(a["logs"], b["logs"])):
            if x != y: print("first differing iteration", i, x, y); break
        print("max dX", np.abs(a["X"] - b["X"]).max())
