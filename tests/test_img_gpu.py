"""Parity of the HIP KeyFrame / Tracker image path (through the C ABI) against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    from mcptam_amd import synth_img
    return synth_img.make_tracking_scene()


def _pair(w, h, **kw):
    from mcptam_amd.keyframe import KeyFrame
    from oracle import OracleKeyFrame
    return KeyFrame(w, h, **kw), OracleKeyFrame(w, h, **kw)


def _assert_lite_equal(g, o):
    for l in range(4):
        assert g.LevelSize(l) == o.LevelSize(l)
        assert np.array_equal(g.Image(l), o.Image(l)), "pyramid level %d differs" % l
        assert g.FastThresh(l) == o.FastThresh(l)
        assert np.array_equal(g.FastFrequency(l), o.FastFrequency(l))
        cg, co = g.Corners(l), o.Corners(l)
        assert cg.shape == co.shape and np.array_equal(cg, co), "corner list/order differs at level %d" % l
        assert np.array_equal(g.RowLUT(l), o.RowLUT(l))


@pytest.mark.parametrize("kw", [dict(), dict(pavgb=True), dict(adaptive=False), dict(glare=True)])
def test_make_keyframe_lite_bit_exact(gpu_required, scene, kw):
    g, o = _pair(640, 480, **kw)
    img = scene["imgA"].copy()
    if kw.get("glare"):
        img[100:140, 200:260] = 255          # a saturated blob to mask
    g.MakeKeyFrame_Lite(img)
    o.MakeKeyFrame_Lite(img)
    _assert_lite_equal(g, o)
    assert len(g.Corners(0)) > 500


def test_make_keyframe_lite_with_masks_and_odd_stride(gpu_required, scene):
    g, o = _pair(640, 480)
    rng = np.random.default_rng(5)
    masks = []
    for l in range(4):
        m = np.full((480 >> l, 640 >> l), 255, dtype=np.uint8)
        m[: (100 >> l), :] = 0
        m[rng.integers(0, 480 >> l, 50), rng.integers(0, 640 >> l, 50)] = 254     # "< 255" is masked out too
        masks.append(m if l != 2 else None)
    big = np.zeros((480, 700), dtype=np.uint8)
    big[:, :640] = scene["imgB"]
    view = big[:, :640]                      # row stride 700
    g.MakeKeyFrame_Lite(np.ascontiguousarray(view), masks)
    o.MakeKeyFrame_Lite(np.ascontiguousarray(view), masks)
    _assert_lite_equal(g, o)
    assert (g.Corners(0)[:, 1] >= 100).all()


def test_flat_and_tiny_inputs(gpu_required):
    g, o = _pair(64, 64)
    img = np.full((64, 64), 77, dtype=np.uint8)
    g.MakeKeyFrame_Lite(img)
    o.MakeKeyFrame_Lite(img)
    _assert_lite_equal(g, o)
    assert len(g.Corners(0)) == 0


def test_c5_size_frame(gpu_required):
    """BASELINE config c5 frame size (1280x960)."""
    from mcptam_amd import synth_img
    sc = synth_img.make_tracking_scene(size=(1280, 960))
    g, o = _pair(1280, 960)
    g.MakeKeyFrame_Lite(sc["imgA"])
    o.MakeKeyFrame_Lite(sc["imgA"])
    _assert_lite_equal(g, o)


@pytest.mark.parametrize("use_shi,use_percent,nm", [(False, True, 0), (True, True, 0), (True, False, 0), (False, True, 1)])
def test_make_keyframe_rest_candidates(gpu_required, scene, use_shi, use_percent, nm):
    g, o = _pair(640, 480)
    g.MakeKeyFrame_Lite(scene["imgA"])
    o.MakeKeyFrame_Lite(scene["imgA"])
    g.MakeKeyFrame_Rest(use_shi, use_percent, 0.8, 70.0, nm)
    o.MakeKeyFrame_Rest(use_shi, use_percent, 0.8, 70.0, nm)
    for l in range(4):
        pg, sg = g.Candidates(l)
        po, so = o.Candidates(l)
        assert np.array_equal(pg, po), "candidate set/order differs at level %d" % l
        if use_shi:
            assert np.allclose(sg, so, rtol=1e-12, atol=0)
        else:
            assert np.array_equal(sg, so)
    assert len(g.Candidates(0)[0]) > 100


def test_minipatch_find(gpu_required, scene):
    from mcptam_amd.keyframe import minipatch_find
    from oracle import oracle_minipatch_find
    g, o = _pair(640, 480)
    g2, o2 = _pair(640, 480)
    g.MakeKeyFrame_Lite(scene["imgA"]); o.MakeKeyFrame_Lite(scene["imgA"])
    g2.MakeKeyFrame_Lite(scene["imgB"]); o2.MakeKeyFrame_Lite(scene["imgB"])
    for level, rng in ((0, 10), (1, 20), (2, 5)):
        pos = o.Corners(level)[::7][:300]
        pos = np.concatenate([pos, [[1, 1], [639 >> level, 479 >> level]]]).astype(np.int32)     # border cases
        pg, fg, sg = minipatch_find(g, g2, level, pos, pos, rng)
        po, fo, so = oracle_minipatch_find(o, o2, level, pos, pos, rng)
        assert np.array_equal(fg, fo) and np.array_equal(pg, po) and np.array_equal(sg, so)
    assert fg.sum() > 0


def _points(scene, g, o):
    from mcptam_amd import synth_img
    g.MakeKeyFrame_Rest()
    o.MakeKeyFrame_Rest()
    return synth_img.make_map_points(scene["cam"], g, o, scene["poseA"], scene["depth"])


INT_FIELDS = ("in_image", "search_level", "template_bad", "searched", "found", "did_subpix", "coarse_x", "coarse_y", "score")


def assert_track_equal(og, oo):
    """Integer outputs are bit-exact: template bytes, ZMSSD scores, coarse positions, flags.  (CVD::transform truncates the
    bilinear sample to a byte, so in a flat image region the last ulp of the warped source position decides between grey
    level g and g-1; that position depends on atan(), which both sides now take correctly rounded -- the device in
    double-double arithmetic (csrc/atan_cr.h), the oracle through binary128 -- so there is no tolerance left here.)"""
    assert np.array_equal(og["templ"], oo["templ"]), "template bytes differ in %d of %d templates" % (
        int((og["templ"] != oo["templ"]).any(axis=1).sum()), len(og))
    for f in INT_FIELDS:
        assert np.array_equal(og[f], oo[f]), f
    return 0


@pytest.mark.parametrize("rng,its,exh", [(10, 8, False), (30, 0, False), (5, 3, True)])
def test_track_search_matches_oracle(gpu_required, scene, rng, its, exh):
    from mcptam_amd.keyframe import track_search
    from oracle import oracle_track_search
    gA, oA = _pair(640, 480)
    gB, oB = _pair(640, 480)
    gA.MakeKeyFrame_Lite(scene["imgA"]); oA.MakeKeyFrame_Lite(scene["imgA"])
    gB.MakeKeyFrame_Lite(scene["imgB"]); oB.MakeKeyFrame_Lite(scene["imgB"])
    pts = _points(scene, gA, oA)
    if exh:
        pts = pts[:120]
    pts[3]["fixed"] = 1                                     # calibration-style point: exhaustive + 10 sub-pixel iterations
    pts.append(dict(pts[0], world_pos=np.array([0.0, 0.0, -5.0])))     # behind the camera: not in image
    I = (np.eye(3), np.zeros(3))
    cfb = (np.eye(3), np.array([0.02, 0.0, 0.0]))
    RB, tB = scene["poseB"]
    bfw = (RB, tB - cfb[1])
    og = track_search(gB, scene["cam"], bfw, cfb, pts, rng, its, exh)
    oo = oracle_track_search(oB, scene["cam"], bfw, cfb, pts, rng, its, exh)
    assert_track_equal(og, oo)
    for f in ("image", "cam_derivs", "jacobian", "warp_inverse"):
        assert np.allclose(og[f], oo[f], rtol=1e-11, atol=1e-12), f
    assert np.array_equal(og["sqrt_inv_noise"], oo["sqrt_inv_noise"])
    assert np.allclose(og["found_pos"], oo["found_pos"], rtol=0, atol=1e-9)
    assert og["found"].sum() > 0.5 * len(pts) or exh
    assert og["in_image"][-1] == 0


def test_pose_update_matches_oracle(gpu_required, scene):
    from mcptam_amd.keyframe import track_pose_update, track_search
    from oracle import oracle_track_pose_update
    gA, oA = _pair(640, 480)
    gB, oB = _pair(640, 480)
    gA.MakeKeyFrame_Lite(scene["imgA"]); oA.MakeKeyFrame_Lite(scene["imgA"])
    gB.MakeKeyFrame_Lite(scene["imgB"])
    pts = _points(scene, gA, oA)
    I = (np.eye(3), np.zeros(3))
    RB, tB = scene["poseB"]
    out = track_search(gB, scene["cam"], (RB, tB + np.array([0.01, 0.0, 0.0])), I, pts, 10, 8)
    for override in (-1.0, 16.0):
        mg, wg, sg = track_pose_update(out["found"], out["found_pos"], out["image"], out["sqrt_inv_noise"], out["jacobian"], override)
        mo, wo, so = oracle_track_pose_update(out["found"], out["found_pos"], out["image"], out["sqrt_inv_noise"], out["jacobian"], override)
        assert abs(sg - so) <= 1e-13 * so
        assert np.array_equal(wg == 0, wo == 0)             # same outlier set
        assert np.allclose(wg, wo, rtol=1e-12, atol=0)
        assert np.allclose(mg, mo, rtol=1e-9, atol=1e-13)
    assert abs(mg[0] + 0.01) < 2e-3                          # the update undoes the 1 cm perturbation
    none = np.zeros(len(pts), dtype=np.uint8)
    mg, _, _ = track_pose_update(none, out["found_pos"], out["image"], out["sqrt_inv_noise"], out["jacobian"])
    assert np.all(mg == 0)                                   # no measurements: zero update (Tracker.cc:1421-1422)


def test_epipolar_hypotheses_batch(gpu_required, scene):
    """the map-side PatchFinder caller (MapMakerServerBase::AddPointEpipolar, :724-781) hands one point per
    depth hypothesis to PatchFinder with range 3 and no sub-pixel step -- the same batch entry serves it."""
    from mcptam_amd import synth_img
    from mcptam_amd.keyframe import track_search
    from oracle import oracle_track_search
    gA, oA = _pair(640, 480)
    gB, oB = _pair(640, 480)
    gA.MakeKeyFrame_Lite(scene["imgA"]); oA.MakeKeyFrame_Lite(scene["imgA"])
    gB.MakeKeyFrame_Lite(scene["imgB"]); oB.MakeKeyFrame_Lite(scene["imgB"])
    gA.MakeKeyFrame_Rest(); oA.MakeKeyFrame_Rest()
    cand, _ = gA.Candidates(1)
    cand = cand[::max(1, len(cand)//25)][:25]
    true_scale = []
    pts, owner = [], []
    ray_scales = np.linspace(3.0, 12.0, 31)
    for ci, c in enumerate(cand):
        ray = scene["cam"].unproject(np.array([[(c[0] + 0.5)*2 - 0.5, (c[1] + 0.5)*2 - 0.5]]))[0]
        true_scale.append(scene["depth"]/ray[2])
        for s_ in ray_scales:
            pts.append(synth_img.hypothesis_point(scene["cam"], gA, oA, scene["poseA"], c, 1, s_))
            owner.append(ci)
    I = (np.eye(3), np.zeros(3))
    og = track_search(gB, scene["cam"], scene["poseB"], I, pts, 3, 0)
    oo = oracle_track_search(oB, scene["cam"], scene["poseB"], I, pts, 3, 0)
    assert_track_equal(og, oo)
    owner = np.array(owner)
    good = 0
    for ci in range(len(cand)):
        m = (owner == ci) & (og["found"] == 1)
        if not m.any():
            continue
        idx = np.nonzero(m)[0]
        best = idx[np.argmin(og["score"][idx])]
        k_true = ci*len(ray_scales) + int(np.argmin(np.abs(ray_scales - true_scale[ci])))
        lvl = 1 << int(og["search_level"][best])
        pred = og["image"][k_true]/lvl                      # where the true 3-D point projects, at the search level
        if abs(og["coarse_x"][best] - pred[0]) <= 3.5 and abs(og["coarse_y"][best] - pred[1]) <= 3.5:
            good += 1
    assert good >= 0.6*len(cand)          # the best-scoring hypothesis locks onto the corner of the true 3-D point


def test_gpu_matches_committed_image_fixture(gpu_required):
    """The HIP image path against the committed expected outputs (tests/golden/img_320.npz): bit-exact pyramid, corners,
    row LUT, thresholds and candidates; the tracker batch with the documented flat-region tolerance."""
    import os
    from mcptam_amd import synth_img
    from mcptam_amd.keyframe import KeyFrame, track_search
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "img_320.npz"))
    A, B = KeyFrame(320, 240), KeyFrame(320, 240)
    A.MakeKeyFrame_Lite(g["imgA"])
    B.MakeKeyFrame_Lite(g["imgB"])
    A.MakeKeyFrame_Rest()
    for l in range(4):
        assert np.array_equal(A.Image(l), g["imgA_l%d" % l])
        assert np.array_equal(A.Corners(l), g["cornersA%d" % l])
        assert np.array_equal(A.RowLUT(l), g["lutA%d" % l])
        assert A.FastThresh(l) == int(g["threshA%d" % l])
        assert np.array_equal(A.Candidates(l)[0], g["candA%d" % l])
    sc = synth_img.make_tracking_scene(size=(320, 240))
    assert np.array_equal(sc["imgA"], g["imgA"])                               # the generator is deterministic
    pts = synth_img.make_map_points(sc["cam"], A, A, sc["poseA"], sc["depth"], per_level=(120, 80, 40, 10))
    out = track_search(B, sc["cam"], sc["poseB"], (np.eye(3), np.zeros(3)), pts, 10, 8)
    assert_track_equal(out, g["track"])


def test_candidate_stability_pruning_with_history(gpu_required):
    """MakeKeyFrame_Rest on a handle that has seen earlier frames (the tracker reuses one KeyFrame per camera): the
    previous two frames stay resident (Level::imagePrev / vCornersPrev) and every candidate must survive the
    back-and-forward MiniPatch walk, src/KeyFrame.cc:456-527.  Bit-exact candidate lists after 1, 2 and 3 pushes."""
    from mcptam_amd import synth_img
    g, o = _pair(640, 480)
    sc = synth_img.make_tracking_scene()
    frames = [sc["imgA"], sc["imgB"], np.roll(sc["imgB"], 2, axis=1), np.roll(sc["imgA"], -1, axis=0)]
    survivors = []
    for i, f in enumerate(frames):
        g.MakeKeyFrame_Lite(f)
        o.MakeKeyFrame_Lite(f)
        assert g.NumPrev() == o.NumPrev() == min(i, 2)
        _assert_lite_equal(g, o)
        for kw in (dict(), dict(use_shi=True), dict(use_percent=False, thresh=70.0)):
            g.MakeKeyFrame_Rest(**kw)
            o.MakeKeyFrame_Rest(**kw)
            for l in range(4):
                pg, sg = g.Candidates(l)
                po, so = o.Candidates(l)
                assert np.array_equal(pg, po), (i, kw, l)
                assert np.array_equal(sg, so)
        survivors.append(sum(len(g.Candidates(l)[0]) for l in range(4)))
    assert survivors[1] < survivors[0]            # the pruning removes unstable candidates once there is history
    assert survivors[1] > 20


def _sbi_frames(scene):
    from mcptam_amd import synth_img
    return synth_img.make_smooth_scene()


def test_small_blurry_image_bit_exact(gpu_required, scene):
    """SmallBlurryImage::MakeFromKF + MakeJacs (src/SmallBlurryImage.cc:67-118): thumbnail bytes, float template and
    gradient image identical to the oracle (integer resize, float Gaussian with the same tap order, contraction off)."""
    for img in _sbi_frames(scene):
        for blur in (2.5, 1.0):
            g, o = _pair(640, 480)
            g.MakeKeyFrame_Lite(img); o.MakeKeyFrame_Lite(img)
            g.MakeSBI(blur); o.MakeSBI(blur)
            sg, tg, jg = g.SBI()
            so, to, jo = o.SBI()
            assert np.array_equal(sg, so)
            assert np.array_equal(tg, to), np.abs(tg - to).max()
            assert np.array_equal(jg, jo)
            assert abs(float(tg.mean())) < 0.1*float(tg.std()) and tg.std() > 1.0


def test_relocaliser_scores_and_esm_alignment(gpu_required, scene):
    """Relocaliser::ScoreKFs (bit-exact ZMSSD, first-smallest winner), IteratePosRelToTarget (ESM SE2, double sums in a
    different but fixed order: 1e-9 relative) and SE3fromSE2, src/Relocaliser.cc:61-121, src/SmallBlurryImage.cc:139-310."""
    from mcptam_amd.keyframe import sbi_iterate, sbi_score, sbi_se3_from_se2
    from mcptam_amd.taylor_camera import TaylorCamera
    from oracle import oracle_sbi_iterate, oracle_sbi_score, oracle_sbi_se3_from_se2
    frames = _sbi_frames(scene)
    G, O = [], []
    for img in frames:
        g, o = _pair(640, 480)
        g.MakeKeyFrame_Lite(img); o.MakeKeyFrame_Lite(img)
        g.MakeSBI(); o.MakeSBI()
        G.append(g); O.append(o)
    nosbi_g, nosbi_o = _pair(640, 480)
    nosbi_g.MakeKeyFrame_Lite(frames[0]); nosbi_o.MakeKeyFrame_Lite(frames[0])
    bg, sg = sbi_score(G[1], [G[2], nosbi_g, G[0], G[1], G[0]])
    bo, so = oracle_sbi_score(O[1], [O[2], nosbi_o, O[0], O[1], O[0]])
    assert bg == bo == 3 and np.array_equal(sg, so)                      # itself; the SBI-less keyframe is skipped (DBL_MAX)
    bg, sg = sbi_score(G[1], [G[2], G[0], G[0]])
    assert bg == 1 and sg[1] == sg[2] < sg[0]                            # ties: the first one wins
    cam = TaylorCamera(scene["cam"].params, (640, 480), (640, 480), (40, 30))
    for its in (1, 6, 10):
        Rg, tg, scg = sbi_iterate(G[1], G[0], its)
        Ro, to, sco = oracle_sbi_iterate(O[1], O[0], its)
        assert np.allclose(Rg, Ro, rtol=0, atol=1e-9) and np.allclose(tg, to, rtol=0, atol=1e-8) and abs(scg - sco) <= 1e-8*sco
        R3g = sbi_se3_from_se2(Rg, tg, cam, cam)
        R3o = oracle_sbi_se3_from_se2(Ro, to, cam, cam)
        assert np.allclose(R3g, R3o, rtol=0, atol=1e-9) and np.allclose(R3g @ R3g.T, np.eye(3), atol=1e-12)
    ang = np.degrees(np.arctan2(Rg[1, 0], Rg[0, 0]))
    assert 2.5 < abs(ang) < 3.8                                          # the 3 degree in-plane rotation is found
    Rg, tg, scg = sbi_iterate(G[0], G[0], 6)
    assert np.array_equal(Rg, np.eye(2)) and np.array_equal(tg, np.zeros(2)) and scg == 0.0


def test_sbi_rotation_against_last_frame(gpu_required, scene):
    """Tracker::CalcSBIRotation's per-camera step: MakeSBI keeps the previous SBI of the handle as 'last frame'."""
    from mcptam_amd.keyframe import sbi_iterate
    frames = _sbi_frames(scene)
    a, b = _pair(640, 480)[0], _pair(640, 480)[0]
    cur = _pair(640, 480)[0]
    a.MakeKeyFrame_Lite(frames[0]); a.MakeSBI()
    b.MakeKeyFrame_Lite(frames[1]); b.MakeSBI()
    cur.MakeKeyFrame_Lite(frames[0]); cur.MakeSBI()
    with pytest.raises(RuntimeError):
        cur.SBIRotationFromLast()
    cur.MakeKeyFrame_Lite(frames[1]); cur.MakeSBI()
    R1, t1, s1 = cur.SBIRotationFromLast(6)
    R2, t2, s2 = sbi_iterate(b, a, 6)
    assert np.array_equal(R1, R2) and np.array_equal(t1, t2) and s1 == s2


@pytest.mark.parametrize("w,h", [(642, 482), (322, 250), (64, 64)])
def test_odd_sizes_and_degenerate_frames(gpu_required, w, h):
    """Ragged / empty inputs: dimensions that do not halve evenly (level sizes are size/2 rounded down, KeyFrame.cc:189),
    a black frame (no corners anywhere: empty lists, zero LUT, no candidates, nothing found), a saturated frame with
    glare masking, a mask that removes everything, and a noise frame -- all identical to the oracle."""
    rng = np.random.default_rng(w*1000 + h)
    noise = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    blobs = np.zeros((h, w), dtype=np.uint8)
    for _ in range(60):
        x, y = int(rng.integers(4, w - 12)), int(rng.integers(4, h - 12))
        blobs[y:y + int(rng.integers(3, 9)), x:x + int(rng.integers(3, 9))] = int(rng.integers(60, 255))
    frames = [np.zeros((h, w), dtype=np.uint8), np.full((h, w), 255, dtype=np.uint8), noise, blobs]
    for kw in (dict(), dict(glare=True)):
        g, o = _pair(w, h, **kw)
        for f in frames:
            g.MakeKeyFrame_Lite(f); o.MakeKeyFrame_Lite(f)
            _assert_lite_equal(g, o)
            g.MakeKeyFrame_Rest(); o.MakeKeyFrame_Rest()
            for l in range(4):
                assert np.array_equal(g.Candidates(l)[0], o.Candidates(l)[0])
    g, o = _pair(w, h)
    zero_masks = [np.zeros((h >> l, w >> l), dtype=np.uint8) for l in range(4)]
    g.MakeKeyFrame_Lite(blobs, zero_masks); o.MakeKeyFrame_Lite(blobs, zero_masks)
    _assert_lite_equal(g, o)
    assert all(len(g.Corners(l)) == 0 for l in range(4))
    g.MakeKeyFrame_Lite(frames[0]); o.MakeKeyFrame_Lite(frames[0])
    assert all(len(g.Corners(l)) == 0 for l in range(4)) and all(g.LevelSize(l) == (w >> l, h >> l) for l in range(4))
    if w >= 320:
        g.MakeSBI(); o.MakeSBI()
        assert all(np.array_equal(a, b) for a, b in zip(g.SBI(), o.SBI()))


def test_pose_refine_matches_oracle(gpu_required):
    """mcp_track_pose_refine: the ten Gauss-Newton pose iterations of Tracker::TrackMap in one launch (re-projection at
    iterations 0, 4, 9, linear updates in between, Tukey sigma from the exact median, 16.0 override beyond iteration 5)
    against the oracle: pose, last update, weights, final image positions; also a coarse-stage schedule and no-found input."""
    import test_oracle_cpu as toc
    from mcptam_amd.keyframe import track_pose_refine
    from oracle import oracle_track_pose_refine
    cam, cfbs, bfw, recs = toc._refine_scene()
    f = recs["found"] != 0
    schedules = [(None, None), (np.ones(10, dtype=np.uint8), np.full(10, 1.0)), (np.array([1, 0, 1], dtype=np.uint8), np.zeros(3))]
    for nl, ov in schedules:
        kw = {} if nl is None else dict(nonlinear=nl, override_sigma=ov)
        pg, mg, wg, og = track_pose_refine(recs, [cam, cam], cfbs, bfw, **kw)
        po, mo, wo, oo = oracle_track_pose_refine(recs, [cam, cam], cfbs, bfw, **kw)
        assert np.allclose(pg[0], po[0], rtol=0, atol=1e-10) and np.allclose(pg[1], po[1], rtol=0, atol=1e-10)
        assert np.allclose(mg, mo, rtol=0, atol=1e-10)
        assert np.allclose(wg, wo, rtol=0, atol=1e-8) and np.array_equal(wg == 0, wo == 0)
        assert np.allclose(og["image"][f], oo["image"][f], rtol=0, atol=1e-8) and np.allclose(og["cam_derivs"][f], oo["cam_derivs"][f], rtol=1e-10, atol=1e-8)
        assert np.array_equal(og["image"][~f], recs["image"][~f])
    none = recs.copy(); none["found"] = 0
    pg, mg, wg, og = track_pose_refine(none, [cam, cam], cfbs, bfw)
    assert np.all(mg == 0) and np.array_equal(pg[0], bfw[0]) and np.all(wg == 0)


def test_c5_frame_size_1280x960(gpu_required):
    """BASELINE config c5 (8-camera 1280x960 rig, one camera per GPU): one camera's frame through the whole per-frame path at
    that size -- pyramid, corners, LUT, candidates bit-exact; the tracked-point batch bit-exact in its integer outputs; pose iterations."""
    from mcptam_amd import synth_img
    from mcptam_amd.keyframe import pose_points, track_pose_refine, track_search
    from oracle import oracle_track_pose_refine, oracle_track_search
    sc = synth_img.make_tracking_scene(size=(1280, 960))
    gA, oA = _pair(1280, 960)
    gB, oB = _pair(1280, 960)
    gA.MakeKeyFrame_Lite(sc["imgA"]); oA.MakeKeyFrame_Lite(sc["imgA"])
    gB.MakeKeyFrame_Lite(sc["imgB"]); oB.MakeKeyFrame_Lite(sc["imgB"])
    _assert_lite_equal(gA, oA)
    _assert_lite_equal(gB, oB)
    assert len(gA.Corners(0)) > 2000 and gA.LevelSize(3) == (160, 120)
    gA.MakeKeyFrame_Rest(); oA.MakeKeyFrame_Rest()
    for l in range(4):
        assert np.array_equal(gA.Candidates(l)[0], oA.Candidates(l)[0])
    from mcptam_amd import synth_img as si
    pts = si.make_map_points(sc["cam"], gA, oA, sc["poseA"], sc["depth"], per_level=(500, 300, 150, 50))
    I = (np.eye(3), np.zeros(3))
    og = track_search(gB, sc["cam"], sc["poseB"], I, pts, 10, 8)
    oo = oracle_track_search(oB, sc["cam"], sc["poseB"], I, pts, 10, 8)
    assert_track_equal(og, oo)
    assert og["found"].sum() > 0.2*len(pts)       # twice the pixel motion of the 640x480 scene at the same search radius
    wp = np.array([p["world_pos"] for p in pts])
    pg, mg, wg, _ = track_pose_refine(pose_points(wp, og, 0), [sc["cam"]], [I], sc["poseB"])
    po, mo, wo, _ = oracle_track_pose_refine(pose_points(wp, oo, 0), [sc["cam"]], [I], sc["poseB"])
    assert np.allclose(pg[0], po[0], atol=1e-9) and np.allclose(pg[1], po[1], atol=1e-9)


def test_track_search_with_newton_fallback_camera(gpu_required, scene):
    """The tracker's projection with a camera that has no inverse polynomial (n_inv == 0, TaylorCamera.cc:258-270)."""
    from mcptam_amd.keyframe import track_search
    from mcptam_amd.taylor_camera import TaylorCamera
    from oracle import oracle_track_search
    cam = TaylorCamera(scene["cam"].params, (640, 480), (640, 480), (640, 480), force_newton=True)
    assert cam.to_struct().n_inv == 0
    gA, oA = _pair(640, 480)
    gB, oB = _pair(640, 480)
    gA.MakeKeyFrame_Lite(scene["imgA"]); oA.MakeKeyFrame_Lite(scene["imgA"])
    gB.MakeKeyFrame_Lite(scene["imgB"]); oB.MakeKeyFrame_Lite(scene["imgB"])
    pts = _points(scene, gA, oA)[:300]
    I = (np.eye(3), np.zeros(3))
    og = track_search(gB, cam, scene["poseB"], I, pts, 10, 8, False)
    oo = oracle_track_search(oB, cam, scene["poseB"], I, pts, 10, 8, False)
    assert_track_equal(og, oo)
    assert og["found"].sum() > 100


def test_camera_per_rank_pose_refine(gpu_required):
    """BASELINE config c5's exchange: one camera per rank, per pose iteration an all-reduce of every rank's squared
    errors (exact global Tukey median) and of the 6x6 + 6 WLS accumulator (src/Tracker.cc:1386-1512).  (a) one rank through the
    sharded entry = the fused single-launch kernel; (b) two ranks (processes sharing this GPU, gloo transport), camera 0 / camera 1,
    end on the same pose as the single-device refinement of all points, and both ranks hold identical poses."""
    import os
    import socket
    import tempfile
    import torch.multiprocessing as mp
    import dist_workers
    import test_oracle_cpu as toc
    from mcptam_amd.keyframe import track_pose_refine, track_pose_refine_sharded
    cam, cfbs, bfw, recs = toc._refine_scene()
    assert set(np.unique(recs["cam"])) == {0, 1}
    pg, mg, wg, og = track_pose_refine(recs, [cam, cam], cfbs, bfw)
    p1, m1, w1, o1 = track_pose_refine_sharded(recs, [cam, cam], cfbs, bfw)
    assert np.allclose(p1[0], pg[0], rtol=0, atol=1e-13) and np.allclose(p1[1], pg[1], rtol=0, atol=1e-13)
    assert np.allclose(m1, mg, rtol=0, atol=1e-13) and np.allclose(w1, wg, rtol=0, atol=1e-12)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(dist_workers.sharded_pose_refine, args=(2, port, d), nprocs=2, join=True)
        r0, r1 = np.load(os.path.join(d, "refine_0.npz")), np.load(os.path.join(d, "refine_1.npz"))
    assert np.array_equal(r0["R"], r1["R"]) and np.array_equal(r0["t"], r1["t"]) and np.array_equal(r0["mu"], r1["mu"])
    assert np.allclose(r0["R"], pg[0], rtol=0, atol=1e-11) and np.allclose(r0["t"], pg[1], rtol=0, atol=1e-11)
    assert np.allclose(r0["w"], wg[recs["cam"] == 0], rtol=0, atol=1e-9) and np.allclose(r1["w"], wg[recs["cam"] == 1], rtol=0, atol=1e-9)
    assert int(r0["calls"]) == 20                       # two collectives per iteration, ten iterations


def test_c5_tracker_frames_and_window_ba_pipelined(gpu_required):
    """BASELINE config c5's pipeline on one device: the per-frame tracker path (1280x960 pyramid + FAST + PatchFinder search + pose
    iterations) keeps running while a windowed local bundle (BundleAdjustRecent shape) is being adjusted by another host thread --
    every handle has its own HIP stream, nothing is shared.  Both must give exactly what they give alone."""
    import threading
    from mcptam_amd import synth, synth_img
    from mcptam_amd.chain_bundle import ChainBundle
    from mcptam_amd.keyframe import pose_points, track_pose_refine, track_search
    from helpers import run_bundle
    sc = synth_img.make_tracking_scene(size=(1280, 960))
    gA, oA = _pair(1280, 960)
    gA.MakeKeyFrame_Lite(sc["imgA"]); oA.MakeKeyFrame_Lite(sc["imgA"])
    gA.MakeKeyFrame_Rest(); oA.MakeKeyFrame_Rest()
    pts = synth_img.make_map_points(sc["cam"], gA, oA, sc["poseA"], sc["depth"], per_level=(300, 200, 100, 40))
    wp = np.array([p["world_pos"] for p in pts])
    I = (np.eye(3), np.zeros(3))
    from mcptam_amd.keyframe import KeyFrame

    def frame(kf):
        kf.MakeKeyFrame_Lite(sc["imgB"])
        out = track_search(kf, sc["cam"], sc["poseB"], I, pts, 10, 8)
        pose, mu, w, _ = track_pose_refine(pose_points(wp, out, 0), [sc["cam"]], [I], sc["poseB"])
        return out, pose, mu

    cur = KeyFrame(1280, 960)
    ref_out, ref_pose, ref_mu = frame(cur)
    win = synth.recent_window(synth.make_config("c2", n_mkf=40, n_points=6000))
    ref_ba = run_bundle(ChainBundle(win.cams, True, True, False), win, 10)

    result = {}

    def ba_thread():
        result["ba"] = [run_bundle(ChainBundle(win.cams, True, True, False), win, 10) for _ in range(3)]
    th = threading.Thread(target=ba_thread)
    th.start()
    frames = []
    while th.is_alive() or len(frames) < 5:
        frames.append(frame(cur))
        if len(frames) > 400:
            break
    th.join()
    assert len(frames) >= 5
    for out, pose, mu in frames:
        assert np.array_equal(out["templ"], ref_out["templ"]) and np.array_equal(out["found"], ref_out["found"])
        assert np.array_equal(out["found_pos"], ref_out["found_pos"])
        assert np.array_equal(pose[0], ref_pose[0]) and np.array_equal(pose[1], ref_pose[1]) and np.array_equal(mu, ref_mu)
    for r in result["ba"]:
        assert r["rc"] == ref_ba["rc"] and r["logs"] == ref_ba["logs"]
        assert np.array_equal(r["R"], ref_ba["R"]) and np.array_equal(r["t"], ref_ba["t"]) and np.array_equal(r["X"], ref_ba["X"])


def test_frame_batch_equals_per_camera_calls(gpu_required, scene):
    """mcp_kf_make_lite_batch / mcp_track_search_batch: the cameras of a frame in one submission (different image sizes, one
    camera masked, second frame so that the history rotates) give exactly what per-camera calls and the oracle give."""
    from mcptam_amd import hip_rt
    from mcptam_amd.keyframe import KeyFrame, make_lite_batch, track_search, track_search_batch
    from oracle import OracleKeyFrame
    sizes = [(640, 480), (642, 482), (322, 250), (640, 480)]
    rng = np.random.default_rng(11)
    imgs = []
    for (w, h) in sizes:
        base = np.zeros((h, w), dtype=np.uint8)
        src = scene["imgA"] if len(imgs) != 3 else scene["imgB"]
        base[:min(h, 480), :min(w, 640)] = src[:min(h, 480), :min(w, 640)]
        imgs.append(base)
    masks = [None, None, None, []]
    for l in range(4):
        m = np.full((480 >> l, 640 >> l), 255, dtype=np.uint8)
        m[:, : (200 >> l)] = 0
        masks[3].append(m)
    kfs = [KeyFrame(w, h) for (w, h) in sizes]
    oks = [OracleKeyFrame(w, h) for (w, h) in sizes]
    for rep in range(2):
        frame = [np.roll(a, 3*rep, axis=1) for a in imgs]
        make_lite_batch(kfs, frame, masks)
        for c, o in enumerate(oks):
            o.MakeKeyFrame_Lite(frame[c], masks[c])
            _assert_lite_equal(kfs[c], o)
    assert kfs[0].NumPrev() == 1
    assert (kfs[3].Corners(0)[:, 0] >= 200).all()
    # the same frame from a device-resident ring (row stride 704), nothing uploaded by the call
    ring = []
    for c, (w, h) in enumerate(sizes):
        padded = np.zeros((h, 704), dtype=np.uint8)
        padded[:, :w] = np.roll(imgs[c], 3, axis=1)
        ring.append(hip_rt.dev_alloc(padded.nbytes))
        hip_rt.dev_upload(ring[c], padded)
    kd = [KeyFrame(w, h) for (w, h) in sizes]
    make_lite_batch(kd, ring, masks, on_device=True, strides=[704]*4)
    for c in range(4):
        _assert_lite_equal(kd[c], oks[c])
    for r in ring:
        hip_rt.dev_free(r)
    # batched search == per-camera search
    gA, oA = _pair(640, 480)
    gA.MakeKeyFrame_Lite(scene["imgA"]); oA.MakeKeyFrame_Lite(scene["imgA"])
    pts = _points(scene, gA, oA)
    tg = [KeyFrame(640, 480) for _ in range(3)]
    make_lite_batch(tg, [scene["imgB"], scene["imgA"], scene["imgB"]])
    cfbs = [(np.eye(3), np.array([0.02*c, 0.0, 0.0])) for c in range(3)]
    RB, tB = scene["poseB"]
    bfw = (RB, tB)
    lists = [pts, pts[:37], []]
    outs = track_search_batch(tg, [scene["cam"]]*3, bfw, cfbs, lists, 10, 8)
    for c in range(2):
        single = track_search(tg[c], scene["cam"], bfw, cfbs[c], lists[c], 10, 8)
        for f in single.dtype.names:
            assert np.array_equal(single[f], outs[c][f], equal_nan=single[f].dtype.kind == "f"), f
    assert len(outs[2]) == 0 and outs[0]["found"].sum() > 100


def test_frame_batch_rejects_bad_arguments(gpu_required, scene):
    from mcptam_amd.keyframe import KeyFrame, make_lite_batch
    a, b = KeyFrame(640, 480), KeyFrame(640, 480, adaptive=False)
    with pytest.raises(RuntimeError):
        make_lite_batch([a, a], [scene["imgA"], scene["imgA"]])                 # one handle twice
    with pytest.raises(RuntimeError):
        make_lite_batch([a, b], [scene["imgA"], scene["imgA"]])                 # different threshold modes in one launch
    with pytest.raises(RuntimeError):
        make_lite_batch([KeyFrame(64, 64) for _ in range(9)], [np.zeros((64, 64), np.uint8)]*9)      # more than MCP_MAX_FRAME_CAMS
    make_lite_batch([a], [scene["imgA"]])                                       # the handle is still usable afterwards
    assert len(a.Corners(0)) > 500


def test_c5_eight_camera_frame_in_one_submission_beside_window_ba(gpu_required):
    """BASELINE config c5 at its stated size on one device: the eight 1280x960 cameras of a frame go through
    mcp_kf_make_lite_batch (three launches), mcp_track_search_batch (one launch) and the ten pose iterations, frame after frame,
    while another host thread adjusts a BundleAdjustRecent-shaped window.  Every frame must equal the per-camera reference and
    the oracle's pyramid / corners; every window solve must equal the solve run alone."""
    import threading
    from mcptam_amd import synth, synth_img
    from mcptam_amd.chain_bundle import ChainBundle
    from mcptam_amd.keyframe import KeyFrame, make_lite_batch, pose_points, track_pose_refine, track_search, track_search_batch, pack_points
    from helpers import run_bundle
    ncam = 8
    sc = synth_img.make_tracking_scene(size=(1280, 960))
    gA, oA = _pair(1280, 960)
    gA.MakeKeyFrame_Lite(sc["imgA"]); oA.MakeKeyFrame_Lite(sc["imgA"])
    gA.MakeKeyFrame_Rest(); oA.MakeKeyFrame_Rest()
    pts = synth_img.make_map_points(sc["cam"], gA, oA, sc["poseA"], sc["depth"], per_level=(200, 120, 60, 20))
    wp = np.array([p["world_pos"] for p in pts])
    packed = pack_points(pts, lambda kf: kf._h)
    cfbs = [(np.eye(3), np.array([0.01*c, 0.0, 0.0])) for c in range(ncam)]
    # per-camera reference (single-camera entries) and the oracle's view of the frame
    oB = _pair(1280, 960)[1]
    oB.MakeKeyFrame_Lite(sc["imgB"])
    single = KeyFrame(1280, 960)
    single.MakeKeyFrame_Lite(sc["imgB"])
    _assert_lite_equal(single, oB)
    ref_out = [track_search(single, sc["cam"], sc["poseB"], cfbs[c], packed, 10, 8) for c in range(ncam)]
    ref_pose, ref_mu, _, _ = track_pose_refine(np.concatenate([pose_points(wp, ref_out[c], c) for c in range(ncam)]), [sc["cam"]]*ncam, cfbs, sc["poseB"])
    cur = [KeyFrame(1280, 960) for _ in range(ncam)]

    def frame():
        make_lite_batch(cur, [sc["imgB"]]*ncam)
        outs = track_search_batch(cur, [sc["cam"]]*ncam, sc["poseB"], cfbs, [packed]*ncam, 10, 8)
        pose, mu, w, _ = track_pose_refine(np.concatenate([pose_points(wp, outs[c], c) for c in range(ncam)]), [sc["cam"]]*ncam, cfbs, sc["poseB"])
        return outs, pose, mu

    win = synth.recent_window(synth.make_config("c2", n_mkf=40, n_points=6000))
    ref_ba = run_bundle(ChainBundle(win.cams, True, True, False), win, 10)
    result = {}

    def ba_thread():
        result["ba"] = [run_bundle(ChainBundle(win.cams, True, True, False), win, 10) for _ in range(3)]
    th = threading.Thread(target=ba_thread)
    th.start()
    frames = []
    while th.is_alive() or len(frames) < 3:
        frames.append(frame())
        if len(frames) > 200:
            break
    th.join()
    for c in range(ncam):
        _assert_lite_equal(cur[c], oB)
    for outs, pose, mu in frames:
        for c in range(ncam):
            for f in ("templ", "found", "found_pos", "score", "coarse_x", "coarse_y"):
                assert np.array_equal(outs[c][f], ref_out[c][f]), (c, f)
        assert np.array_equal(pose[0], ref_pose[0]) and np.array_equal(pose[1], ref_pose[1]) and np.array_equal(mu, ref_mu)
    assert sum(int(o_["found"].sum()) for o_ in frames[0][0]) > 8*80
    for r in result["ba"]:
        assert r["rc"] == ref_ba["rc"] and r["logs"] == ref_ba["logs"]
        assert np.array_equal(r["R"], ref_ba["R"]) and np.array_equal(r["t"], ref_ba["t"]) and np.array_equal(r["X"], ref_ba["X"])


def test_tracker_frames_and_two_metric_solves_share_the_device_without_a_stall(gpu_required):
    """The reference runs the tracker in the main loop and the map maker in its own thread (/root/reference/src/System.cc:243
    TrackFrame vs src/MapMaker.cc:131 MapMaker::run): their kernels share the device.  The one-launch factorisation spins on
    hand-offs between its workgroups; round 4's form only made progress with every workgroup of the launch resident at once, which
    a busy device does not promise.  Here three things run at once -- tracker frames on this thread, and TWO metric-size solves on
    two other threads (two persistent launches of up to 4 x 112 workgroups each racing for 512 slots) -- and every LM iteration must
    finish without a fallback to the per-step kernels and without a stall (library time of an iteration within 5 ms of the slowest
    iteration of the same solve run alone), with the numbers of the solve run alone."""
    import threading
    from mcptam_amd import synth, synth_img
    from mcptam_amd.chain_bundle import ChainBundle
    from mcptam_amd.keyframe import KeyFrame, make_lite_batch, pose_points, track_pose_refine, track_search_batch, pack_points
    ncam = 4
    sc = synth_img.make_tracking_scene(size=(640, 480))
    gA, oA = _pair(640, 480)
    gA.MakeKeyFrame_Lite(sc["imgA"]); oA.MakeKeyFrame_Lite(sc["imgA"])
    gA.MakeKeyFrame_Rest(); oA.MakeKeyFrame_Rest()
    pts = synth_img.make_map_points(sc["cam"], gA, oA, sc["poseA"], sc["depth"])
    wp = np.array([p["world_pos"] for p in pts])
    packed = pack_points(pts, lambda kf: kf._h)
    cfbs = [(np.eye(3), np.array([0.01*c, 0.0, 0.0])) for c in range(ncam)]
    cur = [KeyFrame(640, 480) for _ in range(ncam)]

    def frame():
        make_lite_batch(cur, [sc["imgB"]]*ncam)
        outs = track_search_batch(cur, [sc["cam"]]*ncam, sc["poseB"], cfbs, [packed]*ncam, 10, 8)
        pose, mu, _, _ = track_pose_refine(np.concatenate([pose_points(wp, outs[c], c) for c in range(ncam)]), [sc["cam"]]*ncam, cfbs, sc["poseB"])
        return pose, mu
    ref_frame = frame()

    p = synth.make_config("metric")
    NIT = 12

    def solve():
        g = ChainBundle(p.cams, True, True, False, disable_convergence=True)
        p.populate(g)
        g.Prepare()
        ms, fb = [], 0
        for _ in range(NIT):                      # (one LM iteration per call: the library's own clock brackets each)
            assert g.Compute(1) == 1
            tm = g.Timing()
            ms.append(tm["total_ms"]); fb += tm["n_persist_fallbacks"]
        R, t = g.GetPoses(p.ids["mkf"])
        logs = g.IterLogs()
        g.close()
        return dict(ms=ms, fallbacks=fb, R=R, t=t, logs=logs)
    alone = solve()
    assert alone["fallbacks"] == 0
    out = {}

    def ba_thread(k):
        out[k] = solve()
    ths = [threading.Thread(target=ba_thread, args=(k,)) for k in range(2)]
    for th in ths:
        th.start()
    nframes = 0
    while any(th.is_alive() for th in ths) and nframes < 2000:
        pose, mu = frame()
        nframes += 1
        assert np.array_equal(pose[0], ref_frame[0][0]) and np.array_equal(pose[1], ref_frame[0][1]) and np.array_equal(mu, ref_frame[1])
    for th in ths:
        th.join()
    assert nframes >= 3
    for k in range(2):
        r = out[k]
        assert r["fallbacks"] == 0
        assert r["logs"] == alone["logs"] and np.array_equal(r["R"], alone["R"]) and np.array_equal(r["t"], alone["t"])
        assert max(r["ms"][1:]) <= max(alone["ms"][1:]) + 5.0, (r["ms"], alone["ms"])


def _assert_states_equal(sg, so):
    for f in ("valid", "point_key", "template_bad", "jacs_valid", "templ", "last_warp"):
        assert np.array_equal(sg[f], so[f]), f
    jv = so["jacs_valid"] == 1
    assert np.array_equal(sg["jac_templ"][jv], so["jac_templ"][jv])
    assert np.allclose(sg["mean_diff"], so["mean_diff"], rtol=0, atol=1e-9)


def _moved(pose, drot, dt):
    from mcptam_amd.synth import so3_exp
    R, t = pose
    return so3_exp(np.asarray(drot, dtype=np.float64)) @ R, np.asarray(t) + np.asarray(dt)


def test_template_cache_over_a_frame_sequence_matches_oracle(gpu_required, scene):
    """PatchFinder::MakeTemplateCoarseCont's cache (src/PatchFinder.cc:144-181) through mcp_patch_sequences, one finder per tracked
    point carried over three frames (creep, creep, jump): outputs AND the finders' members equal the oracle's after every frame;
    in the creep frames the template bytes are frame 1's although a fresh warp would give others."""
    from mcptam_amd import keyframe as kf
    from oracle import oracle_patch_sequences
    gA, oA = _pair(640, 480); gB, oB = _pair(640, 480)
    gA.MakeKeyFrame_Lite(scene["imgA"]); oA.MakeKeyFrame_Lite(scene["imgA"])
    gB.MakeKeyFrame_Lite(scene["imgB"]); oB.MakeKeyFrame_Lite(scene["imgB"])
    pts = _points(scene, gA, oA)
    pts[5]["fixed"] = 1
    I = (np.eye(3), np.zeros(3))
    cam = scene["cam"]
    poses = [scene["poseB"], _moved(scene["poseB"], (0.0004, -0.0003, 0.0015), (0.004, -0.002, 0.003)),
             _moved(scene["poseB"], (0.0006, -0.0002, 0.0025), (0.006, -0.001, 0.005)), _moved(scene["poseB"], (0.01, 0.02, 0.25), (0.05, 0.02, 0.4))]
    seqs = [[dict(point=p, point_key=i, target=0)] for i, p in enumerate(pts)]
    sg, so = kf.new_pf_states(len(pts)), kf.new_pf_states(len(pts))
    first = None
    for fi, pose in enumerate(poses):
        og = kf.patch_sequences(kf.PF_TRACK, [(gB, cam, pose, I)], seqs, sg, 10, 8)
        oo = oracle_patch_sequences(kf.PF_TRACK, [(oB, cam, pose, I)], seqs, so, 10, 8)
        assert_track_equal(og, oo)
        assert np.allclose(og["found_pos"], oo["found_pos"], rtol=0, atol=1e-9)
        _assert_states_equal(sg, so)
        if fi == 0:
            first = og.copy()
        elif fi < 3:
            fresh = kf.track_search(gB, cam, pose, I, pts, 10, 8)
            same = (og["templ"] == first["templ"]).all(axis=1) & (og["search_level"] >= 0) & (first["search_level"] >= 0)
            assert same.sum() > 0.7 * len(pts)
            assert ((fresh["templ"] != og["templ"]).any(axis=1) & same).sum() >= 5, "the cache must be observable"
    fresh = kf.track_search(gB, cam, poses[3], I, pts, 10, 8)
    assert_track_equal(og, fresh)                                      # after the jump every template was remade


def test_map_maker_patchfinder_flows_match_oracle(gpu_required, scene):
    """SURVEY.md 8(f)-1, both map-side callers with the reference's stateful PatchFinder: MapMakerServerBase::ReFind_Common
    (src/MapMakerServerBase.cc:921-1002; one new point walked over several keyframes by ONE finder, sub-pixel only above level 0 and
    kept unconverged) and AddPointEpipolar's two loops (:745-853; one finder and one MapPoint over all depth hypotheses, then
    IterateSubPixToConvergence(10) from the caller's start position on the same finder) -- device against oracle, items and states."""
    from mcptam_amd import keyframe as kf, synth_img
    from oracle import oracle_patch_sequences
    gA, oA = _pair(640, 480); gB, oB = _pair(640, 480); gC, oC = _pair(640, 480)
    gA.MakeKeyFrame_Lite(scene["imgA"]); oA.MakeKeyFrame_Lite(scene["imgA"])
    mask0 = np.full((480, 640), 255, dtype=np.uint8); mask0[:, 400:] = 0               # a static mask on the right part of the target
    gB.MakeKeyFrame_Lite(scene["imgB"], [mask0, None, None, None]); oB.MakeKeyFrame_Lite(scene["imgB"], [mask0, None, None, None])
    gC.MakeKeyFrame_Lite(scene["imgB"]); oC.MakeKeyFrame_Lite(scene["imgB"])
    pts = _points(scene, gA, oA)
    I = (np.eye(3), np.zeros(3))
    cam = scene["cam"]
    poseB = scene["poseB"]
    poseC = _moved(poseB, (0.0003, -0.0002, 0.001), (0.003, -0.001, 0.002))            # a keyframe whose warps differ by < 0.07 from B's
    # ---- ReFindNewlyMade: every point over the keyframes C (pose B), C again seen from a slightly different pose: 2 items per sequence
    tg_g = [(gC, cam, poseB, I), (gC, cam, poseC, I)]
    tg_o = [(oC, cam, poseB, I), (oC, cam, poseC, I)]
    sub = pts[::3]
    seqs = [[dict(point=p, point_key=i, target=0), dict(point=p, point_key=i, target=1)] for i, p in enumerate(sub)]
    sg, so = kf.new_pf_states(len(sub)), kf.new_pf_states(len(sub))
    rg = kf.patch_sequences(kf.PF_REFIND, tg_g, seqs, sg, 4)
    ro = oracle_patch_sequences(kf.PF_REFIND, tg_o, seqs, so, 4)
    assert_track_equal(rg, ro)
    assert np.allclose(rg["found_pos"], ro["found_pos"], rtol=0, atol=1e-9)
    _assert_states_equal(sg, so)
    second = rg[1::2]
    assert ((second["templ"] == rg[0::2]["templ"]).all(axis=1) & (second["search_level"] >= 0)).sum() > 0.7 * len(sub)      # second keyframe: cached template
    up = (rg["found"] == 1) & (rg["search_level"] > 0)
    assert up.any() and (rg["did_subpix"][up] == 1).all() and (rg["did_subpix"][(rg["found"] == 1) & (rg["search_level"] == 0)] == 0).all()
    # ---- AddPointEpipolar: 20 candidates x 45 hypotheses, one finder per candidate; the mask removes the hypotheses projecting right of x = 400
    cand, _ = gA.Candidates(1)
    cand = cand[::max(1, len(cand) // 20)][:20]
    scales = np.concatenate([np.linspace(5.0, 7.0, 41), [7.0], np.linspace(7.0, 7.05, 3)])
    seqs, allh = [], []
    for ci, c in enumerate(cand):
        hyp = [synth_img.hypothesis_point(cam, gA, oA, scene["poseA"], c, 1, s_) for s_ in scales]
        for k in ("pixel_right_w", "pixel_down_w"):
            hyp[41][k] = np.asarray(hyp[41][k]) * 0.3                                  # a hypothesis CalcSearchLevelAndWarpMatrix rejects
        allh.append(hyp)
        seqs.append([dict(point=h, point_key=100 + ci, target=0) for h in hyp])
    sg, so = kf.new_pf_states(len(cand)), kf.new_pf_states(len(cand))
    eg = kf.patch_sequences(kf.PF_EPI_COARSE, [(gB, cam, poseB, I)], seqs, sg, 3)
    eo = oracle_patch_sequences(kf.PF_EPI_COARSE, [(oB, cam, poseB, I)], seqs, so, 3)
    assert_track_equal(eg, eo)
    _assert_states_equal(sg, so)
    eg2 = eg.reshape(len(cand), len(scales))
    assert (eg2["in_image"][eg2["image"][:, :, 0] >= 401] == 0).all() and (eg2["in_image"] == 1).any()       # masked hypotheses are skipped
    live = eg2["in_image"][:, 40] == 1
    assert live.any() and (eg2["template_bad"][live, 42] == 1).all() and (eg2["searched"][live, 42] == 0).all()      # poisoned by the rejected warp before it
    # second loop: the best one to three hypotheses per candidate, in score order, on the same finders
    ref_seqs = []
    for ci in range(len(cand)):
        f = np.nonzero(eg2["found"][ci] == 1)[0]
        order = f[np.argsort(eg2["score"][ci][f], kind="stable")][:3]
        ref_seqs.append([dict(point=allh[ci][j], point_key=100 + ci, target=0, start_pos=eg2["found_pos"][ci][j]) for j in order])
    assert sum(len(s_) for s_ in ref_seqs) >= 10
    fg = kf.patch_sequences(kf.PF_EPI_REFINE, [(gB, cam, poseB, I)], ref_seqs, sg, 3)
    fo = oracle_patch_sequences(kf.PF_EPI_REFINE, [(oB, cam, poseB, I)], ref_seqs, so, 3)
    assert_track_equal(fg, fo)
    assert np.allclose(fg["found_pos"], fo["found_pos"], rtol=0, atol=1e-9)
    _assert_states_equal(sg, so)
    assert (fg["found"] == 1).sum() >= 0.5 * len(fg) and (fg["did_subpix"] == 1).all()


@pytest.mark.parametrize("est", ["Cauchy", "Huber"])
def test_pose_update_and_refine_with_the_other_m_estimators(gpu_required, est):
    """Tracker::CalcPoseUpdate dispatches on Tracker::sMEstimatorName (src/Tracker.cc:1388-1401): Cauchy and Huber next to the default
    Tukey (include/mcptam/MEstimator.h:126-204) -- single update and the ten-iteration loop, both pose-refine kernels, against the oracle."""
    import os
    import test_oracle_cpu as toc
    from mcptam_amd.keyframe import track_pose_update, track_pose_refine
    from oracle import oracle_track_pose_update, oracle_track_pose_refine
    cam, cfbs, bfw, recs = toc._refine_scene()
    rng = np.random.default_rng(3)
    n = len(recs)
    J = rng.normal(size=(n, 12)) * 30
    mg, wg, sg = track_pose_update(recs["found"].astype(np.uint8), recs["found_pos"], recs["image"], recs["sqrt_inv_noise"], J, -1.0, est)
    mo, wo, so = oracle_track_pose_update(recs["found"].astype(np.uint8), recs["found_pos"], recs["image"], recs["sqrt_inv_noise"], J, -1.0, est)
    assert abs(sg - so) <= 1e-13 * so and np.allclose(wg, wo, rtol=1e-12, atol=0) and np.allclose(mg, mo, rtol=1e-9, atol=1e-13)
    pg, mug, wgl, outg = track_pose_refine(recs, [cam, cam], cfbs, bfw, estimator=est)
    po, muo, wol, outo = oracle_track_pose_refine(recs, [cam, cam], cfbs, bfw, estimator=est)
    assert np.allclose(pg[0], po[0], rtol=0, atol=1e-11) and np.allclose(pg[1], po[1], rtol=0, atol=1e-11)
    assert np.allclose(mug, muo, rtol=1e-7, atol=1e-12) and np.allclose(wgl, wol, rtol=1e-9, atol=1e-12)
    pt, _, wt, _ = track_pose_refine(recs, [cam, cam], cfbs, bfw)
    assert np.abs(pt[1] - pg[1]).max() > 1e-9          # another estimator, another pose


def _median_cases(rng, n):
    """(name, error magnitudes): the squared-error populations the Tukey median of the register-held pose iterations has to select from."""
    z = rng.uniform(0.1, 3.0, n); z[rng.random(n) < 0.4] = 0.0
    zz = rng.uniform(0.1, 3.0, n); zz[rng.random(n) < 0.6] = 0.0      # median 0: sigma 0, weights 0/0 in the reference too
    return [("ties", rng.choice([0.5, 1.0, 2.0], n)),
            ("sixty_binades", 2.0**rng.uniform(-30, 30, n)),
            ("all_equal", np.full(n, 0.75)),
            ("many_zero", z), ("zero_median", zz),
            ("lognormal", np.exp(rng.normal(0.0, 1.0, n))),
            ("dense_cluster", 1.0 + np.arange(n)*2.0**-50),
            ("outlier_first", np.concatenate([[2.0**20], rng.uniform(0.5, 1.5, n - 1)])),
            ("tiny_first", np.concatenate([[2.0**-20], rng.uniform(0.5, 1.5, n - 1)]))]


def test_vote_select_unit_against_sort(gpu_required, tmp_path):
    """regs_vote_select (csrc/ba_select.h) alone, in a 512-thread kernel of its own (tests/cpp/vote_select_check.hip): the rank-k key of up to
    1024 register-held keys for k = 0, n/2, n - 1 over eight populations (ties, sixty binades, zeros, a dense cluster ...), with and
    without dead slots, with the window of binades starting below, at and above the answer -- the answer must equal std::sort's whenever the
    window holds it, and the routine must say so when it does not (1944 launches, three calls each on the same LDS tables)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "vote_select_check")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(root, "mcptam_amd", "csrc"),
                           os.path.join(root, "tests", "cpp", "vote_select_check.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("ok "), out.stdout[-2000:] + out.stderr[-500:]


@pytest.mark.parametrize("n", [1000, 1024, 130, 3, 1])
def test_pose_iterations_take_the_exact_median_of_hard_populations(gpu_required, n):
    """k_pose_refine_regs selects Tukey's median (MEstimator.h:194-204: element [size/2] of the sorted squared errors) by votes over a
    window of binades around a guess, falling back to histogram passes (ba_select.h regs_vote_select): ties, populations spread over
    sixty binades, a guess far off (first point an outlier / tiny), many errors exactly zero, a thousand keys that
    differ in their last mantissa bits only, and the unfound gaps in between -- every one against the oracle's sort, after one, two
    and three iterations (the first guess is a sample, the later ones the previous median)."""
    import test_oracle_cpu as toc
    from mcptam_amd.keyframe import track_pose_refine
    from oracle import oracle_track_pose_refine
    cam, cfbs, bfw, recs0 = toc._refine_scene()
    rng = np.random.default_rng(n)
    recs = np.tile(recs0[recs0["found"] != 0], n//8 + 1)[:n].copy()
    for name, err in _median_cases(rng, n):
        r = recs.copy()
        r["sqrt_inv_noise"] = 1.0
        r["found_pos"] = r["image"] + np.stack([err, np.zeros(n)], axis=1)
        if n > 100:
            r["found"][rng.random(n) < 0.15] = 0
        for nit in (1, 2, 3):
            nl = np.array([1, 0, 0][:nit], dtype=np.uint8)
            pg, mg, wg, og = track_pose_refine(r, [cam, cam], cfbs, bfw, nonlinear=nl, override_sigma=np.zeros(nit))
            po, mo, wo, oo = oracle_track_pose_refine(r, [cam, cam], cfbs, bfw, nonlinear=nl, override_sigma=np.zeros(nit))
            ok = np.isfinite(wo).all() and np.isfinite(mo).all()
            if not ok:                                # (a zero median: the call has to come back, with a non-finite update like the reference's)
                assert not np.isfinite(mg).all(), (name, nit)
                continue
            assert np.array_equal(wg == 0, wo == 0), (name, nit)
            assert np.allclose(wg, wo, rtol=1e-9, atol=1e-12), (name, nit, np.abs(wg - wo).max())
            assert np.allclose(mg, mo, rtol=1e-6, atol=1e-9*max(1.0, np.abs(mo).max())), (name, nit)


@pytest.mark.parametrize("est", ["Tukey", "Huber"])
def test_pose_refine_over_many_workgroups_matches_oracle(gpu_required, est, monkeypatch):
    """k_pose_refine_multi: the pose iterations with the points sliced over workgroups (frames with thousands of tracked points,
    BASELINE c5) -- global exact median through device-scope histograms + candidate gather, partial WLS sums added in workgroup
    order, every workgroup solving the same 6 x 6.  Against the oracle on 6000 points in 4 cameras (the default route at that size),
    forced on the small scene (two workgroups), with schedules that exercise every median / override combination, and repeatable
    bit for bit."""
    import test_oracle_cpu as toc
    from mcptam_amd.keyframe import track_pose_refine
    from oracle import oracle_track_pose_refine
    cam, cfbs, bfw, recs = toc._refine_scene()
    rng = np.random.default_rng(17)
    big = np.concatenate([recs]*12)[:6000].copy()
    big["found_pos"] += rng.normal(size=(len(big), 2))*0.3                     # no two points alike
    big["cam"] = rng.integers(0, 2, size=len(big))
    big["found"][rng.uniform(size=len(big)) < 0.1] = 0
    schedules = [(None, None), (np.ones(10, dtype=np.uint8), np.full(10, -1.0)), (np.array([1, 0, 0, 1], dtype=np.uint8), np.array([-1.0, 4.0, -1.0, -1.0]))]
    for nl, ov in schedules:
        kw = dict(estimator=est) if nl is None else dict(nonlinear=nl, override_sigma=ov, estimator=est)
        pg, mg, wg, og = track_pose_refine(big, [cam, cam], cfbs, bfw, **kw)
        po, mo, wo, oo = oracle_track_pose_refine(big, [cam, cam], cfbs, bfw, **kw)
        assert np.allclose(pg[0], po[0], rtol=0, atol=1e-10) and np.allclose(pg[1], po[1], rtol=0, atol=1e-10)
        assert np.allclose(mg, mo, rtol=0, atol=1e-10)
        assert np.allclose(wg, wo, rtol=0, atol=1e-8) and np.array_equal(wg == 0, wo == 0)          # same outlier set: the median is exact
        f = big["found"] != 0
        assert np.allclose(og["image"][f], oo["image"][f], rtol=0, atol=1e-8)
        p2, m2, w2, _ = track_pose_refine(big, [cam, cam], cfbs, bfw, **kw)
        assert np.array_equal(pg[0], p2[0]) and np.array_equal(pg[1], p2[1]) and np.array_equal(mg, m2) and np.array_equal(wg, w2)
    # the same with the median taken from all squared errors gathered into every workgroup's LDS (one barrier instead of three)
    monkeypatch.setenv("MCP_TRACK_REFINE_GATHER", "1")
    ph, mh, wh, _ = track_pose_refine(big, [cam, cam], cfbs, bfw, estimator=est)
    monkeypatch.delenv("MCP_TRACK_REFINE_GATHER")
    pq, mq, wq, _ = track_pose_refine(big, [cam, cam], cfbs, bfw, estimator=est)
    assert np.array_equal(ph[1], pq[1]) and np.array_equal(mh, mq) and np.array_equal(wh, wq)          # an exact median either way: identical runs
    monkeypatch.setenv("MCP_TRACK_REFINE_MULTI", "2")
    pg, mg, wg, og = track_pose_refine(recs, [cam, cam], cfbs, bfw, estimator=est)
    po, mo, wo, oo = oracle_track_pose_refine(recs, [cam, cam], cfbs, bfw, estimator=est)
    assert np.allclose(pg[1], po[1], rtol=0, atol=1e-9) and np.allclose(mg, mo, rtol=0, atol=1e-9) and np.array_equal(wg == 0, wo == 0)
    none = big.copy(); none["found"] = 0
    pg, mg, wg, og = track_pose_refine(none, [cam, cam], cfbs, bfw)
    assert np.all(mg == 0) and np.array_equal(pg[0], bfw[0]) and np.all(wg == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("stateful", [False, True])
def test_track_frame_in_one_submission_equals_the_three_calls(gpu_required, scene, stateful):
    """mcp_track_frame = MakeKeyFrame_Lite of every camera + SearchForPoints + the ten pose iterations of one Tracker::TrackMap stage in one
    submission (src/Tracker.cc:303-318, 985-1075, 1299-1384), the pose points packed on the device.  Same kernels, same data as
    mcp_kf_make_lite_batch + mcp_track_search_batch (or, stateful, mcp_patch_sequences with persistent finders) + host packing +
    mcp_track_pose_refine: every output bit for bit, over two frames (history rotation, template cache), one camera without points;
    the refined pose also against the oracle's iterations."""
    from mcptam_amd import keyframe as kf
    from oracle import oracle_track_pose_update
    cam = scene["cam"]
    gA, oA = _pair(640, 480)
    gA.MakeKeyFrame_Lite(scene["imgA"]); oA.MakeKeyFrame_Lite(scene["imgA"])
    pts = _points(scene, gA, oA)
    lists = [pts, pts[:41], [], pts[7:90]]
    ncam = len(lists)
    cfbs = [(np.eye(3), np.array([0.01*c, 0.0, 0.0])) for c in range(ncam)]
    cfb_arr = np.ascontiguousarray(np.stack([kf._pose12(*c) for c in cfbs]))
    wpos = [np.array([p["world_pos"] for p in l]).reshape(-1, 3) for l in lists]
    fused_kfs = [kf.KeyFrame(640, 480) for _ in range(ncam)]
    plain_kfs = [kf.KeyFrame(640, 480) for _ in range(ncam)]
    tf = kf.TrackFrame(fused_kfs, [cam]*ncam, cfb_arr, lists, stateful=stateful)
    states = [kf.new_pf_states(len(l)) for l in lists]
    frames = [[scene["imgB"]]*ncam, [np.roll(scene["imgB"], 1, axis=1)]*ncam]
    poses = [scene["poseB"], _moved(scene["poseB"], (0.0004, -0.0003, 0.0015), (0.004, -0.002, 0.003))]
    for imgs, pose in zip(frames, poses):
        outs, recs, pose_f, mu_f, w_f = tf.run(imgs, pose, 10, 8)
        kf.make_lite_batch(plain_kfs, imgs)
        if stateful:
            ref = []
            for c in range(ncam):
                seqs = [[dict(point=p, point_key=i, target=0)] for i, p in enumerate(lists[c])]
                ref.append(kf.patch_sequences(kf.PF_TRACK, [(plain_kfs[c], cam, pose, cfbs[c])], seqs, states[c], 10, 8) if lists[c] else np.zeros(0, dtype=kf.TD_OUT_DTYPE))
        else:
            ref = kf.track_search_batch(plain_kfs, [cam]*ncam, pose, cfb_arr, lists, 10, 8)
        for c in range(ncam):
            assert len(outs[c]) == len(lists[c])
            for f in ref[c].dtype.names:
                assert np.array_equal(ref[c][f], outs[c][f], equal_nan=ref[c][f].dtype.kind == "f"), (c, f)
            _assert_lite_equal_gpu = [np.array_equal(fused_kfs[c].Image(l), plain_kfs[c].Image(l)) and np.array_equal(fused_kfs[c].Corners(l), plain_kfs[c].Corners(l)) for l in range(4)]
            assert all(_assert_lite_equal_gpu)
            if stateful and len(lists[c]):
                for f in tf.states[c].dtype.names:
                    assert np.array_equal(tf.states[c][f], states[c][f], equal_nan=tf.states[c][f].dtype.kind == "f"), (c, f)
        host_recs = kf.pose_points_frame(wpos, ref)
        pose_p, mu_p, w_p, recs_p = kf.track_pose_refine(host_recs, [cam]*ncam, cfb_arr, pose)
        assert np.array_equal(pose_f[0], pose_p[0]) and np.array_equal(pose_f[1], pose_p[1]) and np.array_equal(mu_f, mu_p) and np.array_equal(w_f, w_p)
        for f in recs_p.dtype.names:
            assert np.array_equal(recs[f], recs_p[f]), f
        assert sum(int(o_["found"].sum()) for o_ in outs) > 150 and np.abs(mu_f).max() > 0
    # zero-copy results (out = NULL + mcp_track_frame_view, include/mcp_img.h): the same bytes, read in place in the library's pinned block
    if not stateful:
        outs_c, _, pose_c, mu_c, w_c = tf.run(frames[1], poses[1], 10, 8)
        outs_c = [o_.copy() for o_ in outs_c]
        outs_v, _, pose_v, mu_v, w_v = tf.run(frames[1], poses[1], 10, 8, view=True)
        for c in range(ncam):
            assert len(outs_v[c]) == len(lists[c])
            for f in kf.TD_OUT_DTYPE.names:
                assert np.array_equal(outs_c[c][f], outs_v[c][f], equal_nan=outs_c[c][f].dtype.kind == "f"), (c, f)
        assert np.array_equal(pose_c[0], pose_v[0]) and np.array_equal(pose_c[1], pose_v[1]) and np.array_equal(mu_c, mu_v)
    # search only (n_iter = 0): the pose stays, the weights are zero
    outs, recs, pose0, mu0, w0 = tf.run(None, poses[1], 10, 8, nonlinear=np.zeros(0, dtype=np.uint8), override_sigma=np.zeros(0))
    assert np.array_equal(pose0[0], poses[1][0]) and not mu0.any() and not w0.any()


@pytest.mark.timeout(600)
def test_camera_per_rank_frame_loop_runs_on_two_ranks(gpu_required):
    """scripts/bench_tracker.py per_rank_loop: BASELINE c5's shape (a camera per GPU: pyramids + FAST + search on the camera's own rank,
    the pose iterations across the ranks) as a runnable loop -- two ranks on this box's single GPU through gloo; both ranks end every
    frame on the same pose, and it is the single-device pose of the same two cameras."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "scripts", "bench_tracker.py"), "c5", "--gpus", "2", "--frames", "3", "--small", "--debug-single-device"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=500, cwd=root, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    two = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["value"] > 0 and two["found_on_rank_0"] > 100
    out1 = subprocess.run([sys.executable, os.path.join(root, "scripts", "bench_tracker.py"), "c5", "--gpus", "1", "--frames", "2", "--small"],
                          capture_output=True, text=True, timeout=500, cwd=root, env=env)
    assert out1.returncode == 0, out1.stdout[-2000:] + out1.stderr[-2000:]
    one = json.loads([l for l in out1.stdout.splitlines() if l.startswith("{")][-1])
    assert np.abs(np.array(two["pose_t"]) - np.array(one["pose_t"])).max() < 1e-9


def test_pose_iterations_survive_a_workgroup_that_gives_up(gpu_required, monkeypatch):
    """The multi-workgroup pose iterations wait for each other; next to the mapper's kernels not every workgroup need be resident and one
    may give up.  The frame is not lost: the single-workgroup kernel redoes the iterations from a kept copy of the points (forced here
    on every call).  Same poses as the undisturbed run to the rounding of another summation order."""
    import test_oracle_cpu as toc
    from mcptam_amd.keyframe import track_pose_refine
    cam, cfbs, bfw, recs = toc._refine_scene()
    rng = np.random.default_rng(5)
    big = np.concatenate([recs]*12)[:6000].copy()
    big["found_pos"] += rng.normal(size=(len(big), 2))*0.3
    big["cam"] = rng.integers(0, 2, size=len(big))
    pg, mg, wg, og = track_pose_refine(big, [cam, cam], cfbs, bfw)
    monkeypatch.setenv("MCP_TRACK_TEST_PRM_GIVEUP", "1")
    pr, mr, wr, orr = track_pose_refine(big, [cam, cam], cfbs, bfw)
    assert np.allclose(pr[0], pg[0], rtol=0, atol=1e-11) and np.allclose(pr[1], pg[1], rtol=0, atol=1e-11) and np.allclose(mr, mg, rtol=0, atol=1e-11)
    assert np.array_equal(wr == 0, wg == 0) and np.allclose(wr, wg, rtol=0, atol=1e-9)
    assert np.allclose(orr["image"], og["image"], rtol=0, atol=1e-9)
