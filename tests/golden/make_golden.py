"""Regenerates the golden fixtures under tests/golden/.

The reference ships no tests, fixtures or golden vectors for the hot path and cannot be built or
imported here, so these fixtures are produced by THIS repo's CPU oracle from
seeded synthetic inputs.  They pin the oracle against regressions and give the GPU tests an
expected output that does not depend on rebuilding the oracle.  Inputs + expected outputs only.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from helpers import run_bundle  # noqa: E402
from mcptam_amd import synth, synth_img  # noqa: E402
from oracle import OracleBundle, OracleKeyFrame, oracle_track_search  # noqa: E402


def ba_fixture(name, iters, **over):
    p = synth.make_config(name, **over)
    o = OracleBundle(p.cams, True, True, False)
    ids = p.populate(o)
    chi2, err = o.Eval()
    csum, sig = o.DebugRobustChi2()
    r = run_bundle(OracleBundle(p.cams, True, True, False), p, iters)
    logs = np.array([[l["chi2_start"], l["chi2_end"], l["lambda_end"], l["sigma_sq"], l["trials"], l["accepted"]] for l in r["logs"]])
    np.savez_compressed(os.path.join(HERE, "ba_%s.npz" % name), config=name, iters=iters,
                        chi2_init=chi2, robust_chi2_init=csum, sigma_sq_init=sig, rc=r["rc"], logs=logs,
                        R=r["R"], t=r["t"], X=r["X"], outliers=np.array(r["outliers"], dtype=np.int32).reshape(-1, 3),
                        sigma_sq=r["sigma_sq"], mean_chi2=r["mean_chi2"], lam=r["lam"], max_cov=r["max_cov"])


def img_fixture():
    sc = synth_img.make_tracking_scene(size=(320, 240))
    A, B = OracleKeyFrame(320, 240), OracleKeyFrame(320, 240)
    A.MakeKeyFrame_Lite(sc["imgA"])
    B.MakeKeyFrame_Lite(sc["imgB"])
    A.MakeKeyFrame_Rest()
    pts = synth_img.make_map_points(sc["cam"], A, A, sc["poseA"], sc["depth"], per_level=(120, 80, 40, 10))
    out = oracle_track_search(B, sc["cam"], sc["poseB"], (np.eye(3), np.zeros(3)), pts, 10, 8)
    d = dict(imgA=sc["imgA"], imgB=sc["imgB"], track=out)
    for l in range(4):
        d["cornersA%d" % l] = A.Corners(l)
        d["lutA%d" % l] = A.RowLUT(l)
        d["threshA%d" % l] = A.FastThresh(l)
        d["imgA_l%d" % l] = A.Image(l)
        d["candA%d" % l] = A.Candidates(l)[0]
    np.savez_compressed(os.path.join(HERE, "img_320.npz"), **d)


if __name__ == "__main__":
    ba_fixture("tiny", 12)
    ba_fixture("c1", 12)
    ba_fixture("calib", 10)
    img_fixture()
    print("golden fixtures written to", HERE)
