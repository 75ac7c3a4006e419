import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcptam_amd import synth_img
from mcptam_amd.keyframe import KeyFrame, track_search
from oracle import OracleKeyFrame, oracle_track_search
scene = synth_img.make_tracking_scene()
gA, oA, gB, oB = KeyFrame(640,480), OracleKeyFrame(640,480), KeyFrame(640,480), OracleKeyFrame(640,480)
gA.MakeKeyFrame_Lite(scene["imgA"]); oA.MakeKeyFrame_Lite(scene["imgA"]); gB.MakeKeyFrame_Lite(scene["imgB"]); oB.MakeKeyFrame_Lite(scene["imgB"])
gA.MakeKeyFrame_Rest(); oA.MakeKeyFrame_Rest()
cand, _ = gA.Candidates(1); cand = cand[::max(1, len(cand)//25)][:25]
pts = []
for c in cand:
    for s_ in np.linspace(3.0, 12.0, 31):
        pts.append(synth_img.hypothesis_point(scene["cam"], gA, oA, scene["poseA"], c, 1, s_))
I = (np.eye(3), np.zeros(3))
og = track_search(gB, scene["cam"], scene["poseB"], I, pts, 3, 0)
oo = oracle_track_search(oB, scene["cam"], scene["poseB"], I, pts, 3, 0)
d = np.nonzero(og["score"] != oo["score"])[0]
print("ndiff", len(d))
for i in d[:6]:
    print(i, "gpu", og["score"][i], og["found"][i], og["coarse_x"][i], og["coarse_y"][i], "orc", oo["score"][i], oo["found"][i], oo["coarse_x"][i], oo["coarse_y"][i], "level", og["search_level"][i], "img", og["image"][i], "tb", og["template_bad"][i], oo["template_bad"][i])
