#!/usr/bin/env python
"""Secondary ChainBundle lines of bench.py: the BASELINE configurations that are not the headline, on ONE MI355X.

  c2             4-cam, 50 MKF, 10k points, 80k measurements (BASELINE configs[1])
  c4             4-cam, 500 MKF, 100k points, 800k measurements, the WHOLE map on one device (configs[3] without its sharding)
  metric_forced_multi  the headline map driven through the multi-rank machine on the native one-rank RCCL communicator
  c4_rank_shard  one rank's eighth of that map (synth.partition(c4, 8, 0): 500 poses replicated, ~12.5k points, ~100k measurements)
                 driven through the whole multi-rank machine on the native one-rank RCCL communicator (MCP_BA_FORCE_MULTI=1): what
                 ONE of the eight ranks of configs[3] executes per LM iteration, minus the time the other seven's bytes take on the wire

Each block: ms per LM iteration (K iterations between two synchronisations, map resident, convergence actions off), the stage table
of a second, event-timed pass, the dominant stage's roofline entry, and the per-iteration collective bytes the multi-rank machine
counted (c4_rank_shard) -- the inputs of DESIGN.md 6's scaling model.  Used by bench.py (`secondary`) and on its own:

  python scripts/bench_secondary.py [c2 c4 c4_rank_shard] [--steps K]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(name, steps=10, warm=6, device=0):
    import numpy as np
    import bench as B
    from mcptam_amd import chain_bundle, synth
    shard = name == "c4_rank_shard"
    forced = name == "metric_forced_multi"       # the headline map through the multi-rank machine on a one-rank communicator: what the machinery itself costs a rank
    cfg = "c4" if name.startswith("c4") else ("metric" if forced else name)
    whole = synth.make_config(cfg)
    problem = synth.partition(whole, 8, 0) if shard else whole
    shard = shard or forced
    comm = None
    old = os.environ.get("MCP_BA_FORCE_MULTI")
    if shard:
        os.environ["MCP_BA_FORCE_MULTI"] = "1"
        comm = chain_bundle.Comm(chain_bundle.comm_unique_id(), 0, 1, device)
    try:
        def fresh(profile=False):
            b = chain_bundle.ChainBundle(problem.cams, True, True, False, disable_convergence=True, device=device, profile=profile)
            problem.populate(b)
            if comm is not None:
                b.SetComm(comm)
            t0 = time.perf_counter()
            b.Prepare()
            return b, (time.perf_counter() - t0) * 1e3
        bw, _ = fresh()
        bw.Compute(warm)
        bw.close()
        b, prep_ms = fresh()
        t0 = time.perf_counter()
        rc = b.Compute(steps)          # (returns with the state read back: the device is idle again)
        dt = time.perf_counter() - t0
        if rc != steps:
            raise RuntimeError("Compute ran %d of %d iterations (%s)" % (rc, steps, chain_bundle.last_error()))
        logs = b.IterLogs()
        tm_run = b.Timing()
        b.close()
        bp, _ = fresh(profile=True)
        bp.Compute(steps)
        tm = bp.Timing()
        bp.close()
    finally:
        if comm is not None:
            comm.close()
        if shard:
            if old is None:
                os.environ.pop("MCP_BA_FORCE_MULTI", None)
            else:
                os.environ["MCP_BA_FORCE_MULTI"] = old
    np_ = 6 * int((~problem.base_fixed).sum())
    roofs = B.stage_rooflines(tm, problem.n_meas, problem.n_points, np_, tm["n_linearize"], tm["n_trials"], tm["n_solves"])
    for r in roofs.values():
        r["traffic"] = None          # (the PMC pass of this round is of the headline workload: nothing to quote for this size)
    dom = max(roofs.items(), key=lambda kv: kv[1]["avg_ms"] * kv[1]["launches"])
    trials = sum(l["trials"] for l in logs)
    out = {"workload": "%s: %d cams, %d MKF (%d free poses, %d unknowns in the reduced system), %d points, %d measurements%s" % (
               name, len(problem.cams), problem.n_mkf, np_ // 6, np_, problem.n_points, problem.n_meas,
               (" -- the headline map on a one-rank RCCL communicator, every collective of the multi-rank path executed (MCP_BA_FORCE_MULTI=1)" if forced else " -- rank 0's share of the c4 map split over 8 ranks, one-rank RCCL communicator, every collective of the multi-rank path executed") if shard else ""),
           "value": steps / dt, "unit": "LM iterations/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "trials_per_iteration": trials / steps,
           "reduced_system_solves": tm["n_solves"], "prepare_ms": prep_ms, "persist_fallbacks": tm_run["n_persist_fallbacks"] + tm["n_persist_fallbacks"], "factorisation_chains": tm_run.get("chol_chains", 0),
           "stages_ms_per_launch": {k: round(v["avg_ms"], 5) for k, v in roofs.items()},
           "stages_ms_total": {k: tm[k] for k in ("eval_ms", "select_ms", "linearize_ms", "schur_ms", "cholesky_ms", "solve_ms", "update_ms")},
           "roofline": {k: dom[1][k] for k in ("bound", "achieved", "peak", "unit", "frac", "avg_ms", "plan_basis") if k in dom[1]} | {"kernel": dom[0], "traffic": None}}
    if "plan_basis" in out["roofline"]:
        # a long trajectory's plan is banded: the dense (6P)^3 / 3 overstates its work several times over -- the plan's own count is the honest basis here
        pb = out["roofline"].pop("plan_basis")
        out["roofline"].update(achieved=pb["achieved"], frac=pb["frac"], flops_basis="block-sparse plan: %.3g flop per factorisation (dense (6P)^3/3 would be %.3g)" % (pb["flops_per_factorisation"], pb["dense_flops"]))
    if shard:
        out["collectives_per_iteration"] = {"main_lane": tm_run["n_collectives_main"] / steps, "speculative_lane": tm_run["n_collectives_spec"] / steps,
                                            "main_lane_bytes": tm_run["collective_bytes_main"] / steps, "speculative_lane_bytes": tm_run["collective_bytes_spec"] / steps}
    return out


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c2", "c4", "c4_rank_shard", "metric_forced_multi"]
    steps = 10
    if "--steps" in sys.argv:
        steps = int(sys.argv[sys.argv.index("--steps") + 1]); names = [n for n in names if n != str(steps)]
    out = {}
    sys.stdout.flush(); real = os.dup(1); os.dup2(2, 1)          # (RCCL prints its version banner to stdout: the result line stays alone there)
    for n in names:
        try:
            out[n] = run(n, steps=steps)
        except Exception as exc:
            out[n] = {"error": repr(exc)}
    sys.stdout.flush()
    os.write(real, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
