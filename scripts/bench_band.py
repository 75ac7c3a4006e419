#!/usr/bin/env python
"""A trajectory that is a band: the headline map's sizes (4 cameras, 200 MKF, 50k points, 400k measurements) on an open arc (or a loop
walked once) whose poses see the points of their neighbours only -- the map MCPTAM builds while it explores -- with the factorisation
as two chains + separator (the default for such a map) and as one chain (MCP_BA_CHOL_CHAINS=1).

  python scripts/bench_band.py [arc|ring|metric_shuffled] [--steps K]
prints one JSON line: ms per LM iteration and the factorisation's stage time either way, the chains the plan was built with.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(shape, steps, warm=6):
    from mcptam_amd import chain_bundle, synth
    if shape == "metric_shuffled":          # the headline map with its MKFs handed over in a random order (as a std::set of pointers would)
        p = synth.shuffle_mkfs(synth.make_config("metric"))
    else:
        p = synth.make_config("ring_metric" if shape == "ring" else "band_metric")
    out = {"workload": "%s: 4 cams, %d MKF, %d points, %d measurements" % (shape, p.n_mkf, p.n_points, p.n_meas)}
    for label, env in (("two_chains", None), ("one_chain", "1")):
        if env is None:
            os.environ.pop("MCP_BA_CHOL_CHAINS", None)
        else:
            os.environ["MCP_BA_CHOL_CHAINS"] = env

        def fresh(profile=False):
            b = chain_bundle.ChainBundle(p.cams, True, True, False, disable_convergence=True, profile=profile)
            p.populate(b)
            t0 = time.perf_counter()
            b.Prepare()
            return b, (time.perf_counter() - t0) * 1e3
        bw, _ = fresh()
        bw.Compute(warm)
        bw.close()
        chain_bundle.struct_cache_clear()
        b, prep_ms = fresh()
        t0 = time.perf_counter()
        rc = b.Compute(steps)
        dt = time.perf_counter() - t0
        assert rc == steps, chain_bundle.last_error()
        logs = b.IterLogs()
        tm = b.Timing()
        b.close()
        bp, _ = fresh(profile=True)
        bp.Compute(steps)
        tp = bp.Timing()
        bp.close()
        out[label] = {"ms_per_iter": dt * 1e3 / steps, "iters_per_s": steps / dt, "cold_prepare_ms": prep_ms, "chol_chains": tm["chol_chains"],
                      "cholesky_stage_ms": tp["cholesky_ms"] / max(tp["n_solves"], 1), "n_solves": tp["n_solves"],
                      "trials": sum(l["trials"] for l in logs), "chi2_end": logs[-1]["chi2_end"], "persist_fallbacks": tm["n_persist_fallbacks"]}
    os.environ.pop("MCP_BA_CHOL_CHAINS", None)
    return out


if __name__ == "__main__":
    shape = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "arc"
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 20
    print(json.dumps(run(shape, steps)))
