import sys, os, time
sys.path.insert(0, os.getcwd())
from mcptam_amd import chain_bundle, synth
p = synth.make_config("metric")
for rep in range(3):
    chain_bundle.struct_cache_clear()
    b = chain_bundle.ChainBundle(p.cams, True, True, False, disable_convergence=True)
    p.populate(b)
    if rep == 2: os.environ["MCP_BA_TRACE"] = "1"
    t0 = time.perf_counter(); b.Prepare(); t1 = time.perf_counter()
    print("prepare cold %.3f ms  populate(lib) %.3f ms" % ((t1 - t0)*1e3, b.abi_seconds*1e3), flush=True)
    b.close()
