#!/usr/bin/env python
"""What "parity unpinned" can cost: every third-party semantic the oracle restates from memory ([3P-memory]) has an
oracle-only switch; this script runs the metric bundle adjustment (first K iterations) and a tracker frame (BASELINE c3 shape at
320x240 per camera for speed) under each alternative and tabulates which outputs move and by how much.  When a dump of a real MCPTAM
run arrives through mcptam_amd/map_io.py, the first mismatch against the default column says which row to look at.

CPU only (no GPU, no product code):   python scripts/oracle_sensitivity.py [--config metric] [--iters 8] > profiles/r03/oracle_sensitivity.md
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def ba_runs(config, iters):
    from mcptam_amd import synth
    from oracle import OracleBundle
    from helpers import run_bundle, rel_err
    p = synth.make_config(config)
    variants = [
        ("default (g2o as restated: tau 1e-5, +1e-3, lambda*=ni / ni*=2, max(1/3, min(1-(2rho-1)^3, 2/3)))", []),
        ("initial lambda tau = 1e-3 (ba_oracle.c computeLambdaInit)", [(0, 1e-3)]),
        ("initial lambda tau = 1e-7", [(0, 1e-7)]),
        ("rho denominator without the +1e-3", [(1, 0.0)]),
        ("rejection rule lambda *= 2 (no growing ni)", [(2, 1)]),
        ("acceptance rule lambda *= 1/3 always", [(3, 1)]),
        ("acceptance rule without the 2/3 cap", [(3, 2)]),
        ("TaylorCamera::Project with the platform's libm atan (what the reference calls) instead of the correctly rounded one", [("atan_libm", 1)]),
    ]
    rows, base = [], None
    for name, sw in variants:
        o = OracleBundle(p.cams, True, True, False)
        o.DisableConvergence(True)
        try:
            o.SetSolver(2, min(8, os.cpu_count() or 1))     # Schur + OpenMP: same numbers to ~1e-12, minutes instead of an hour
        except Exception:
            pass
        from oracle import img_lib
        atan_libm = any(k == "atan_libm" for k, _ in sw)
        for k, v in sw:
            if k != "atan_libm":
                o.SetVariant(k, v)
        img_lib().orc_set_atan_libm(1 if atan_libm else 0)          # (process-wide oracle switch: the BA oracle and the tracker oracle share orc_atan)
        t0 = time.time()
        try:
            r = run_bundle(o, p, iters)
        finally:
            img_lib().orc_set_atan_libm(0)
        dt = time.time() - t0
        if base is None:
            base = r
        trials = [l["trials"] for l in r["logs"]]
        flips = next((i for i, (a, b) in enumerate(zip(r["logs"], base["logs"])) if a["trials"] != b["trials"] or a["accepted"] != b["accepted"]), None)
        rows.append((name, " ".join(map(str, trials)), "%.6e" % r["logs"][-1]["chi2_end"], "%.3e" % r["lam"],
                     "-" if r is base else "%.2e" % rel_err(r["t"], base["t"]), "-" if r is base else "%.2e" % rel_err(r["X"], base["X"]),
                     "-" if r is base else ("none" if flips is None else "iteration %d" % flips), "%.0f s" % dt))
    return p, rows


def calib_rows(iters=10):
    from mcptam_amd import synth
    from oracle import OracleBundle
    from helpers import run_bundle, rel_err
    p = synth.make_config("calib")
    out = []
    runs = {}
    for sym in (0, 1):
        o = OracleBundle(p.cams, True, True, False)
        o.SetDupSymmetric(sym)
        runs[sym] = run_bundle(o, p, iters)
    conv = {}
    for sym in (0, 1):
        o = OracleBundle(p.cams, True, True, False)
        o.SetDupSymmetric(sym)
        conv[sym] = run_bundle(o, p)
    out.append(("calib map, %d iterations: one-sided (g2o, default) vs symmetric duplicate-vertex block" % iters,
                "%.2e" % rel_err(runs[1]["t"], runs[0]["t"]), "%.2e" % rel_err(runs[1]["X"], runs[0]["X"]),
                "%d vs %d iterations to convergence, final poses agree to %.1e" % (conv[0]["rc"], conv[1]["rc"], rel_err(conv[1]["t"], conv[0]["t"]))))
    return out


def tracker_rows():
    from mcptam_amd import synth_img
    from oracle import OracleKeyFrame, oracle_track_search, img_lib
    sc = synth_img.make_tracking_scene(size=(640, 480))
    cam = sc["cam"]
    I = (np.eye(3), np.zeros(3))
    rows, base = [], None
    L = img_lib()
    for name, kw, nonmax, tround, alibm in (("default (halfSample truncating mean, fast_nonmax on the FAST-10 score, transform truncating, correctly rounded atan)", {}, 0, 0, 0),
                                     ("halfSample = cascaded pavgb (libCVD SSE2 byte path)", {"pavgb": True}, 0, 0, 0),
                                     ("fast_nonmax on the ring-SAD corner_score", {}, 1, 0, 0),
                                     ("CVD::transform byte conversion rounding half up", {}, 0, 1, 0),
                                     ("TaylorCamera::Project with the platform's libm atan (glibc here) instead of the correctly rounded one", {}, 0, 0, 1)):
        L.orc_img_set_variant(0, tround)
        L.orc_set_atan_libm(alibm)
        A, B = OracleKeyFrame(640, 480, **kw), OracleKeyFrame(640, 480, **kw)
        A.MakeKeyFrame_Lite(sc["imgA"]); B.MakeKeyFrame_Lite(sc["imgB"])
        A.MakeKeyFrame_Rest(nonmax_score=nonmax)
        pts = synth_img.make_map_points(cam, A, A, sc["poseA"], sc["depth"])
        out = oracle_track_search(B, cam, sc["poseB"], I, pts, 10, 8)
        corners = [len(B.Corners(l)) for l in range(4)]
        cands = [len(A.Candidates(l)[0]) for l in range(4)]
        rec = dict(corners=corners, cands=cands, npts=len(pts), found=int(out["found"].sum()), out=out, img=[B.Image(l) for l in range(4)])
        if base is None:
            base = rec
        px = [int((rec["img"][l] != base["img"][l]).sum()) for l in range(4)]
        n = min(len(out), len(base["out"]))
        same_pts = rec["cands"] == base["cands"]
        tb = int((out["templ"][:n] != base["out"]["templ"][:n]).any(axis=1).sum()) if same_pts else -1
        fp = float(np.abs(out["found_pos"][:n] - base["out"]["found_pos"][:n])[(out["found"][:n] == 1) & (base["out"]["found"][:n] == 1)].max()) if same_pts else float("nan")
        rows.append((name, " ".join(map(str, px)), " ".join(map(str, corners)), " ".join(map(str, cands)),
                     "%d / %d" % (rec["found"], rec["npts"]), "n/a (other candidates)" if tb < 0 else str(tb), "n/a" if not same_pts else "%.3f" % fp))
    L.orc_img_set_variant(0, 0)
    L.orc_set_atan_libm(0)
    return rows


def table(header, rows):
    print("| " + " | ".join(header) + " |")
    print("|" + "---|" * len(header))
    for r in rows:
        print("| " + " | ".join(str(c) for c in r) + " |")
    print()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="metric")
    ap.add_argument("--iters", type=int, default=8)
    args = ap.parse_args()
    print("# Oracle sensitivity to the [3P-memory] semantics (scripts/oracle_sensitivity.py)\n")
    print("Every row flips ONE restated third-party convention in the CPU oracle (oracle-only switches; the product library implements the "
          "default column) and reports how far the outputs move.  Parity stays *unpinned* whatever this table says; it bounds what being wrong "
          "about a convention would cost, and tells a future cross-check against a real MCPTAM dump where to look first.\n")
    p, rows = ba_runs(args.config, args.iters)
    print("## ChainBundle LM schedule -- `%s` map (%d MKF, %d points, %d measurements), first %d iterations\n" % (args.config, p.n_mkf, p.n_points, p.n_meas, args.iters))
    table(["variant", "trials per iteration", "chi2 after the last iteration", "lambda", "pose t rel. diff", "points rel. diff", "first accept/reject flip", "CPU time"], rows)
    print("## Duplicated pose vertex in one edge (BundleAdjusterCalib shapes, non-fixed points)\n")
    table(["case", "pose t rel. diff", "points rel. diff", "to convergence"], calib_rows())
    print("## Image path -- one 640x480 frame pair, tracker search range 10 with 8 sub-pixel iterations\n")
    table(["variant", "pyramid pixels that differ (L0..L3)", "FAST corners kept (L0..L3)", "candidates (L0..L3)", "points found", "templates with a different byte", "max |found_pos| diff (px)"], tracker_rows())


if __name__ == "__main__":
    main()
