import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_oracle_cpu as toc
from mcptam_amd.keyframe import track_pose_refine, FINE_NONLINEAR, FINE_OVERRIDE
from oracle import oracle_track_pose_refine
cam, cfbs, bfw, recs = toc._refine_scene()
print("n", len(recs), "found", int((recs["found"] != 0).sum()))
for mode in ("0", "2"):
    os.environ["MCP_TRACK_REFINE_MULTI"] = mode
    for k in range(1, 11):
        nl, ov = np.asarray(FINE_NONLINEAR)[:k], np.asarray(FINE_OVERRIDE)[:k]
        pg, mg, wg, og = track_pose_refine(recs, [cam, cam], cfbs, bfw, nonlinear=nl, override_sigma=ov)
        po, mo, wo, oo = oracle_track_pose_refine(recs, [cam, cam], cfbs, bfw, nonlinear=nl, override_sigma=ov)
        print(mode, k, "pose dt %.2e  mu d %.2e  w d %.2e  zero-set equal %s" % (np.abs(pg[1] - po[1]).max(), np.abs(mg - mo).max(), np.abs(wg - wo).max(), np.array_equal(wg == 0, wo == 0)))
