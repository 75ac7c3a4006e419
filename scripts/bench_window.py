"""BundleAdjustRecent window of the metric map (src/BundleAdjusterBase.cc:188-265), whole calls the way BundleAdjusterMulti::BundleAdjust
runs them (fresh handle, bulk replay, Prepare, Compute(10), read-back): per-part host times and the library's own stage times.
Usage: python scripts/bench_window.py [--calls 40] [--config metric]"""
import argparse
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcptam_amd import chain_bundle, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=40)
    ap.add_argument("--config", default="metric")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--profile", action="store_true", help="one more call with event-bracketed stages (library_timing_profiled; no speculation or mailbox in that call)")
    ap.add_argument("--trace", action="store_true", help="one more call at the end with MCP_BA_TRACE=1 MCP_BA_EVT=1 (set-up phases and the device timeline on stderr)")
    args = ap.parse_args()
    problem = synth.make_config(args.config)
    w = synth.recent_window(problem)
    parts = {"create_ms": [], "populate_in_library_ms": [], "prepare_ms": [], "compute_ms": [], "readback_ms": [], "readback_in_library_ms": [], "close_ms": [], "call_ms": []}
    tim = None
    for rep in range(args.calls + 3 + (1 if args.trace else 0)):
        if rep == args.calls + 3:
            os.environ["MCP_BA_TRACE"] = "1"; os.environ["MCP_BA_EVT"] = "1"
        t0 = time.perf_counter()
        b = chain_bundle.ChainBundle(w.cams, True, True, False, device=0)
        ids = w.populate(b)
        t1 = time.perf_counter()
        b.Prepare()
        t2 = time.perf_counter()
        rc = b.Compute(args.iters)
        t3 = time.perf_counter()
        b.GetPoses(ids["mkf"]); b.GetPoints(ids["point"]); b.GetOutlierMeasurements()
        t4 = time.perf_counter()
        tim = b.Timing()
        abi = b.abi_seconds; abi_r = b.abi_read_seconds; abi_c = b.abi_create_seconds
        b.close()
        t5 = time.perf_counter()
        if 3 <= rep < args.calls + 3:
            parts["populate_in_library_ms"].append(abi * 1e3); parts["prepare_ms"].append((t2 - t1) * 1e3)
            parts["compute_ms"].append((t3 - t2) * 1e3); parts["readback_ms"].append((t4 - t3) * 1e3); parts["close_ms"].append((t5 - t4) * 1e3)
            parts["readback_in_library_ms"].append(abi_r * 1e3)
            parts["create_ms"].append(abi_c * 1e3)
            parts["call_ms"].append((abi_c + abi + (t3 - t1) + abi_r + (t5 - t4)) * 1e3)        # what a native caller pays: the library's time in Add*, Prepare, Compute, Get*, destroy
    prof = None
    if args.profile:
        b = chain_bundle.ChainBundle(w.cams, True, True, False, device=0, profile=True)
        w.populate(b); b.Prepare(); b.Compute(args.iters)
        prof = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in b.Timing().items()}
        b.close()
    out = {"workload": "%d MKF (%d free), %d points, %d measurements, %d LM iterations per call" % (w.n_mkf, int((~w.base_fixed).sum()), w.n_points, w.n_meas, args.iters),
           "iterations_run": rc, "ms_median": {k: round(float(np.median(v)), 4) for k, v in parts.items()},
           "ms_min": {k: round(float(np.min(v)), 4) for k, v in parts.items()}, "library_timing_last_call": tim, "library_timing_profiled": prof}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
