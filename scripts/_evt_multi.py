import os, sys
sys.path.insert(0, os.getcwd())
os.environ["MCP_BA_FORCE_MULTI"] = "1"; os.environ["MCP_BA_EVT"] = "1"
from mcptam_amd import chain_bundle, synth
p = synth.make_config("metric")
comm = chain_bundle.Comm(chain_bundle.comm_unique_id(), 0, 1, 0)
b = chain_bundle.ChainBundle(p.cams, True, True, False, disable_convergence=True)
p.populate(b); b.SetComm(comm); b.Prepare()
b.Compute(12)
b.close(); comm.close()
