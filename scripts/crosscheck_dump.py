#!/usr/bin/env python
"""One-command cross-check against a REAL MCPTAM run -- the only route from "parity unpinned" to pinned.

    python scripts/crosscheck_dump.py before.dump after.dump cameras.dump [--iters N] [--fixed-mkf 1] [--tol 1e-6] [--no-gpu]

`before.dump` / `after.dump`: the map as MapMakerBase::DumpToFile writes it (/root/reference/src/MapMakerBase.cc:475-577) right
before and right after ONE global bundle adjustment of the reference (BundleAdjusterMulti::BundleAdjust through
MapMakerServerBase::BundleAdjustAll); `cameras.dump`: SystemBase::DumpCamerasToFile (src/SystemBase.cc:166-215).  `--iters`: the outer
iterations the reference ran (ChainBundle::Compute's return value, printed at ROS_DEBUG; default: run to convergence like the
reference does).

The script replays `before` through (a) the HIP path (C ABI, libmcptam_hip.so) and (b) the CPU oracle in its default configuration
and under every [3P-memory] switch of scripts/oracle_sensitivity.py, and compares the adjusted poses and points with `after`,
element-wise relative (|a - b| <= tol * max(|b|, 1e-3 max|b|)).  It prints one JSON report: per candidate the largest pose / point
deviation from the reference's result, whether it is within `tol`, and -- if the default is not -- which switch explains the dump best.
Exit status 0 = the default oracle AND the HIP path reproduce the reference within `tol`.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

# (name, [(switch, value), ...]) -- the oracle-only switches of orc_ba_set_variant (oracle/ba_oracle.c), as oracle_sensitivity.py lists them
VARIANTS = [
    ("default (g2o as restated in SURVEY.md A.5)", []),
    ("initial lambda tau = 1e-3", [(0, 1e-3)]),
    ("initial lambda tau = 1e-7", [(0, 1e-7)]),
    ("rho denominator without the +1e-3", [(1, 0.0)]),
    ("rejection rule lambda *= 2 (no growing ni)", [(2, 1)]),
    ("acceptance rule lambda *= 1/3 always", [(3, 1)]),
    ("acceptance rule without the 2/3 cap", [(3, 2)]),
]


def elem_rel(a, b, floor_frac=1e-3):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    floor = max(floor_frac * float(np.abs(b).max()), 1e-300)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def adjusted_state(bundle, problem, iters):
    """Replay `problem`, adjust, return (BaseFromWorld R, t per MKF, world position per point, outer iterations run)."""
    from helpers import collect
    ids = problem.populate(bundle)
    rc = bundle.Compute(iters) if iters is not None else bundle.Compute()
    R, t, X = collect(bundle, ids)
    # points live in their source camera's frame: back to world coordinates with the ADJUSTED source pose (BundleAdjusterMulti.cc:298-337)
    k, c = problem.pt_src[:, 0], problem.pt_src[:, 1]
    Rs = np.einsum("nij,njk->nik", problem.cam_R[c], R[k])
    ts = np.einsum("nij,nj->ni", problem.cam_R[c], t[k]) + problem.cam_t[c]
    world = np.einsum("nji,nj->ni", Rs, X - ts)
    return R, t, world, rc


def compare(state, ref):
    R, t, world, rc = state
    return {"iterations": int(rc), "pose_R": elem_rel(R, ref[0]), "pose_t": elem_rel(t, ref[1]), "points": elem_rel(world, ref[2])}


def crosscheck(before, after, cameras, iters=None, n_fixed_mkf=1, tol=1e-6, use_gpu=True):
    from mcptam_amd import map_io
    from oracle import OracleBundle
    cams = map_io.load_cameras(cameras)
    mb, ma = map_io.load_map(before), map_io.load_map(after)
    if len(mb.mkf_pos) != len(ma.mkf_pos) or len(mb.pt_world) != len(ma.pt_world):
        raise SystemExit("the two dumps do not hold the same MKFs / points (outliers removed in between? dump right after the adjustment, "
                         "before HandleBadPoints): %d/%d MKFs, %d/%d points" % (len(mb.mkf_pos), len(ma.mkf_pos), len(mb.pt_world), len(ma.pt_world)))
    p = map_io.problem_from_map(mb, cams, n_fixed_mkf=n_fixed_mkf)
    pa = map_io.problem_from_map(ma, cams, n_fixed_mkf=n_fixed_mkf)
    ref = (pa.base_R, pa.base_t, ma.pt_world)
    # how far did the reference's own adjustment move things?  (a dump pair that did not move cannot discriminate anything)
    moved = {"pose_t": elem_rel(p.base_t, pa.base_t), "points": elem_rel(mb.pt_world, ma.pt_world)}
    report = {"map": {"mkf": int(p.n_mkf), "points": int(p.n_points), "measurements": int(p.n_meas), "cameras": list(mb.cam_names)},
              "reference_adjustment_moved": moved, "tolerance": tol, "dump_precision_note": "DumpToFile prints 6 significant digits by default: "
              "a dump written with the stock precision cannot pin anything below ~1e-6 relative; raise the stream precision in the reference for the cross-check",
              "candidates": []}
    for name, sw in VARIANTS:
        o = OracleBundle(p.cams, True, True, False)
        for k, v in sw:
            o.SetVariant(k, v)
        c = compare(adjusted_state(o, p, iters), ref)
        c["candidate"] = "oracle: " + name
        c["within_tolerance"] = bool(max(c["pose_R"], c["pose_t"], c["points"]) <= tol)
        report["candidates"].append(c)
    if use_gpu:
        from mcptam_amd import chain_bundle
        if chain_bundle.device_count() > 0:
            g = chain_bundle.ChainBundle(p.cams, True, True, False)
            c = compare(adjusted_state(g, p, iters), ref)
            c["candidate"] = "HIP path (libmcptam_hip.so)"
            c["within_tolerance"] = bool(max(c["pose_R"], c["pose_t"], c["points"]) <= tol)
            report["candidates"].append(c)
        else:
            report["hip_path"] = "no gfx950 device here: " + chain_bundle.last_error()
    worst = lambda c: max(c["pose_R"], c["pose_t"], c["points"])
    default = report["candidates"][0]
    best = min(report["candidates"][:len(VARIANTS)], key=worst)
    report["default_oracle_reproduces_the_reference"] = default["within_tolerance"]
    report["best_explaining_oracle_variant"] = best["candidate"]
    if not default["within_tolerance"]:
        report["verdict"] = ("the default restatement does NOT reproduce this adjustment (worst deviation %.2e); the variant closest to the dump is '%s' (%.2e)%s"
                             % (worst(default), best["candidate"], worst(best), "" if best["within_tolerance"] else
                                " -- none of the known switches explains it: look at the residual / Jacobian conventions next (SURVEY.md Appendix A)"))
    else:
        report["verdict"] = "the default restatement reproduces this adjustment within %.0e" % tol
    hip = [c for c in report["candidates"] if c["candidate"].startswith("HIP")]
    report["ok"] = bool(default["within_tolerance"] and (not hip or hip[0]["within_tolerance"]))
    return report


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("before"); ap.add_argument("after"); ap.add_argument("cameras")
    ap.add_argument("--iters", type=int, default=None); ap.add_argument("--fixed-mkf", type=int, default=1)
    ap.add_argument("--tol", type=float, default=1e-6); ap.add_argument("--no-gpu", action="store_true")
    a = ap.parse_args()
    rep = crosscheck(a.before, a.after, a.cameras, iters=a.iters, n_fixed_mkf=a.fixed_mkf, tol=a.tol, use_gpu=not a.no_gpu)
    print(json.dumps(rep, indent=1))
    sys.exit(0 if rep["ok"] else 1)


if __name__ == "__main__":
    main()
