"""ctypes binding of the image-path C ABI (include/mcp_img.h) with the reference's vocabulary:
KeyFrame.MakeKeyFrame_Lite / MakeKeyFrame_Rest, MiniPatch FindPatch, the Tracker's per-point
PatchFinder search and CalcPoseUpdate.  All numeric work happens in libmcptam_hip.so on the GPU."""
import ctypes

import numpy as np

from . import chain_bundle as _cb
from .taylor_camera import McpCamera

LEVELS = 4
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_ubyte_p = ctypes.POINTER(ctypes.c_ubyte)


class McpKfParams(ctypes.Structure):
    _fields_ = [("adaptive_thresh", ctypes.c_int), ("glare_masking", ctypes.c_int), ("half_sample_pavgb", ctypes.c_int),
                ("device", ctypes.c_int)]


class TdIn(ctypes.Structure):
    _fields_ = [("world_pos", ctypes.c_double * 3), ("pixel_right_w", ctypes.c_double * 3), ("pixel_down_w", ctypes.c_double * 3),
                ("source_kf", ctypes.c_void_p), ("source_level", ctypes.c_int), ("center_x", ctypes.c_int),
                ("center_y", ctypes.c_int), ("fixed", ctypes.c_int)]


class TdOut(ctypes.Structure):
    _fields_ = [("image", ctypes.c_double * 2), ("cam_derivs", ctypes.c_double * 4), ("jacobian", ctypes.c_double * 12),
                ("found_pos", ctypes.c_double * 2), ("sqrt_inv_noise", ctypes.c_double), ("warp_inverse", ctypes.c_double * 4),
                ("in_image", ctypes.c_int), ("search_level", ctypes.c_int), ("template_bad", ctypes.c_int),
                ("searched", ctypes.c_int), ("found", ctypes.c_int), ("did_subpix", ctypes.c_int),
                ("coarse_x", ctypes.c_int), ("coarse_y", ctypes.c_int), ("score", ctypes.c_int), ("templ", ctypes.c_ubyte * 64)]


TD_OUT_DTYPE = np.dtype([("image", "f8", 2), ("cam_derivs", "f8", 4), ("jacobian", "f8", 12), ("found_pos", "f8", 2),
                         ("sqrt_inv_noise", "f8"), ("warp_inverse", "f8", 4), ("in_image", "i4"), ("search_level", "i4"),
                         ("template_bad", "i4"), ("searched", "i4"), ("found", "i4"), ("did_subpix", "i4"),
                         ("coarse_x", "i4"), ("coarse_y", "i4"), ("score", "i4"), ("templ", "u1", 64)], align=True)
assert TD_OUT_DTYPE.itemsize == ctypes.sizeof(TdOut)

IMG_SYMBOLS = [
    "mcp_kf_create", "mcp_kf_destroy", "mcp_kf_make_lite", "mcp_kf_make_lite_batch", "mcp_track_search_batch", "mcp_kf_level_size", "mcp_kf_get_image", "mcp_kf_num_corners",
    "mcp_kf_get_corners", "mcp_kf_get_row_lut", "mcp_kf_fast_thresh", "mcp_kf_get_fast_frequency", "mcp_kf_make_rest",
    "mcp_kf_num_prev", "mcp_kf_num_candidates", "mcp_kf_get_candidates", "mcp_minipatch_find", "mcp_track_search", "mcp_track_pose_update",
    "mcp_patch_sequences", "mcp_track_frame", "mcp_track_frame_view", "mcp_track_pose_update_m", "mcp_track_pose_refine_m", "mcp_track_pose_refine_sharded_m", "mcp_track_pose_refine", "mcp_track_pose_refine_sharded", "mcp_kf_make_sbi", "mcp_kf_get_sbi", "mcp_sbi_score", "mcp_sbi_iterate", "mcp_sbi_iterate_last", "mcp_sbi_se3_from_se2",
]
_BOUND = False


def lib():
    global _BOUND
    L = _cb.lib()
    if not _BOUND:
        L.mcp_kf_create.restype = ctypes.c_void_p
        L.mcp_kf_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.mcp_kf_destroy.argtypes = [ctypes.c_void_p]
        L.mcp_kf_make_lite.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.mcp_kf_make_lite_batch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.mcp_track_search_batch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.mcp_kf_level_size.argtypes = [ctypes.c_void_p, ctypes.c_int, c_int_p, c_int_p]
        L.mcp_kf_get_image.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.mcp_kf_num_corners.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mcp_kf_num_prev.argtypes = [ctypes.c_void_p]
        L.mcp_track_pose_refine.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.mcp_track_pose_refine_sharded.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    _cb.ALLREDUCE_FN, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.mcp_kf_make_sbi.argtypes = [ctypes.c_void_p, ctypes.c_double]
        L.mcp_kf_get_sbi.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.mcp_sbi_score.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.mcp_sbi_iterate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.mcp_sbi_iterate_last.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.mcp_sbi_se3_from_se2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.mcp_kf_get_corners.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.mcp_kf_get_row_lut.argtypes = [ctypes.c_void_p, ctypes.c_int, c_int_p]
        L.mcp_kf_fast_thresh.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mcp_kf_get_fast_frequency.argtypes = [ctypes.c_void_p, ctypes.c_int, c_double_p]
        L.mcp_kf_make_rest.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]
        L.mcp_kf_num_candidates.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mcp_kf_get_candidates.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, c_double_p, ctypes.c_int]
        L.mcp_minipatch_find.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.mcp_patch_sequences.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.mcp_track_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.mcp_track_frame.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.mcp_track_pose_update.argtypes = [ctypes.c_int, ctypes.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                            ctypes.c_double, c_double_p, c_double_p, c_double_p]
        _BOUND = True
    return L


def _chk(rc, what):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise RuntimeError("%s failed: %s" % (what, _cb.last_error()))
    return rc


def _dp(a):
    return a.ctypes.data_as(c_double_p)


class KeyFrame:
    """One camera's KeyFrame with a device-resident 4-level pyramid (include/mcptam/KeyFrame.h:93-150)."""

    def __init__(self, w, h, adaptive=True, glare=False, pavgb=False, device=-1):
        self._L = lib()
        prm = McpKfParams(int(adaptive), int(glare), int(pavgb), int(device))
        self._h = self._L.mcp_kf_create(int(w), int(h), ctypes.byref(prm))
        if not self._h:
            raise RuntimeError("mcp_kf_create failed: " + _cb.last_error())
        self.w, self.h = w, h

    def close(self):
        if getattr(self, "_h", None):
            self._L.mcp_kf_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def MakeKeyFrame_Lite(self, img, masks=None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        assert img.shape == (self.h, self.w)
        mp = None
        if masks is not None:
            self._masks = [None if m is None else np.ascontiguousarray(m, dtype=np.uint8) for m in masks]
            arr = (ctypes.c_void_p * LEVELS)(*[None if m is None else m.ctypes.data for m in self._masks])
            mp = ctypes.cast(arr, ctypes.c_void_p)
        _chk(self._L.mcp_kf_make_lite(self._h, img.ctypes.data, img.strides[0], mp), "MakeKeyFrame_Lite")

    # ---- SmallBlurryImage (src/SmallBlurryImage.cc), KeyFrame::MakeSBI (src/KeyFrame.cc:539-545)
    def MakeSBI(self, blur=2.5):
        _chk(self._L.mcp_kf_make_sbi(self._h, float(blur)), "MakeSBI")

    def SBI(self):
        """(mimSmall u8 30x40, mimTemplate f32 30x40, mimImageJacs f32 30x40x2)"""
        small = np.zeros((30, 40), dtype=np.uint8)
        templ = np.zeros((30, 40), dtype=np.float32)
        jacs = np.zeros((30, 40, 2), dtype=np.float32)
        _chk(self._L.mcp_kf_get_sbi(self._h, small.ctypes.data, templ.ctypes.data, jacs.ctypes.data), "SBI")
        return small, templ, jacs

    def SBIRotationFromLast(self, iterations=6):
        """IteratePosRelToTarget of this frame's SBI against the previous one on this handle (Tracker::CalcSBIRotation)."""
        se2 = np.zeros(6)
        sc = np.zeros(1)
        _chk(self._L.mcp_sbi_iterate_last(self._h, int(iterations), se2.ctypes.data, sc.ctypes.data), "SBIRotationFromLast")
        return se2[:4].reshape(2, 2).copy(), se2[4:].copy(), float(sc[0])

    def NumPrev(self):
        """Frames held in the Level::imagePrev / vCornersPrev history (0..2)."""
        return int(self._L.mcp_kf_num_prev(self._h))

    def LevelSize(self, level):
        w, h = ctypes.c_int(), ctypes.c_int()
        _chk(self._L.mcp_kf_level_size(self._h, level, ctypes.byref(w), ctypes.byref(h)), "LevelSize")
        return w.value, h.value

    def Image(self, level):
        w, h = self.LevelSize(level)
        out = np.zeros((h, w), dtype=np.uint8)
        _chk(self._L.mcp_kf_get_image(self._h, level, out.ctypes.data), "Image")
        return out

    def Corners(self, level):
        n = _chk(self._L.mcp_kf_num_corners(self._h, level), "Corners")
        out = np.zeros((max(n, 1), 2), dtype=np.int32)
        n = _chk(self._L.mcp_kf_get_corners(self._h, level, out.ctypes.data, n), "Corners")
        return out[:n]

    def RowLUT(self, level):
        _, h = self.LevelSize(level)
        out = np.zeros(h, dtype=np.int32)
        _chk(self._L.mcp_kf_get_row_lut(self._h, level, out.ctypes.data_as(c_int_p)), "RowLUT")
        return out

    def FastThresh(self, level):
        return _chk(self._L.mcp_kf_fast_thresh(self._h, level), "FastThresh")

    def FastFrequency(self, level):
        out = np.zeros(31)
        _chk(self._L.mcp_kf_get_fast_frequency(self._h, level, _dp(out)), "FastFrequency")
        return out

    def MakeKeyFrame_Rest(self, use_shi=False, use_percent=True, top_fraction=0.8, thresh=70.0, nonmax_score=0):
        _chk(self._L.mcp_kf_make_rest(self._h, int(use_shi), int(use_percent), float(top_fraction), float(thresh), int(nonmax_score)), "MakeKeyFrame_Rest")

    def Candidates(self, level):
        n = _chk(self._L.mcp_kf_num_candidates(self._h, level), "Candidates")
        pos = np.zeros((max(n, 1), 2), dtype=np.int32)
        sc = np.zeros(max(n, 1))
        n = _chk(self._L.mcp_kf_get_candidates(self._h, level, pos.ctypes.data, _dp(sc), n), "Candidates")
        return pos[:n], sc[:n]


def make_lite_batch(kfs, imgs, masks=None, on_device=False, strides=None):
    """MakeKeyFrame_Lite of all cameras of a frame in one submission (the loop of Tracker::TrackFrame, src/Tracker.cc:303-318).
    imgs: one (h, w) uint8 array per camera, or -- on_device=True -- one device address per camera (an image ring that already
    lives in HBM; `strides` = their row strides in bytes, default w)."""
    n = len(kfs)
    hs = (ctypes.c_void_p * n)(*[k._h for k in kfs])
    if on_device:
        ip = (ctypes.c_void_p * n)(*[int(a) for a in imgs])
        st = (ctypes.c_int * n)(*[int(s_) for s_ in (strides or [k.w for k in kfs])])
        keep = None
    else:
        keep = [np.ascontiguousarray(a, dtype=np.uint8) for a in imgs]
        for k, a in zip(kfs, keep):
            assert a.shape == (k.h, k.w)
        ip = (ctypes.c_void_p * n)(*[a.ctypes.data for a in keep])
        st = (ctypes.c_int * n)(*[a.strides[0] for a in keep])
    mp = None
    if masks is not None:
        mkeep, rows = [], []
        for m in masks:
            if m is None:
                rows.append(None)
                continue
            lv = [None if q is None else np.ascontiguousarray(q, dtype=np.uint8) for q in m]
            mkeep.append(lv)
            rows.append((ctypes.c_void_p * LEVELS)(*[None if q is None else q.ctypes.data for q in lv]))
        mp = (ctypes.c_void_p * n)(*[None if r is None else ctypes.cast(r, ctypes.c_void_p) for r in rows])
    _chk(lib().mcp_kf_make_lite_batch(n, hs, ip, st, int(on_device), mp), "make_lite_batch")


def track_search_batch(targets, cams, base_from_world, cams_from_base, points, rng, subpix_its, exhaustive=False):
    """SearchForPoints for all cameras of a frame in one launch; points[c]: list of point dicts or a packed ctypes array."""
    n = len(targets)
    arrs = [p if isinstance(p, ctypes.Array) else pack_points(p, lambda kf: kf._h) for p in points]
    lens = [len(a) for a in arrs]
    whole = np.zeros(sum(lens), dtype=TD_OUT_DTYPE)              # one array, one device-to-host copy; the per-camera results are its slices
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(int)
    outs = [whole[offs[c]:offs[c + 1]] for c in range(n)]
    hs = (ctypes.c_void_p * n)(*[t._h for t in targets])
    cs = cams if isinstance(cams, ctypes.Array) else (type(cams[0].to_struct()) * n)(*[c.to_struct() for c in cams])
    b = _pose12(*base_from_world)
    cfb = np.ascontiguousarray(cams_from_base.reshape(-1)) if isinstance(cams_from_base, np.ndarray) else np.ascontiguousarray(np.concatenate([_pose12(*c) for c in cams_from_base]))
    ns = (ctypes.c_int * n)(*[len(a) for a in arrs])
    ins = (ctypes.c_void_p * n)(*[ctypes.cast(a, ctypes.c_void_p) for a in arrs])
    ops = (ctypes.c_void_p * n)(*[whole.ctypes.data + int(offs[c])*TD_OUT_DTYPE.itemsize for c in range(n)])
    _chk(lib().mcp_track_search_batch(n, hs, cs, _dp(b), _dp(cfb), ns, ins, int(rng), int(subpix_its), int(exhaustive), ops), "track_search_batch")
    return outs


def minipatch_find(src, dst, level, src_pos, dst_pos, rng):
    src_pos = np.ascontiguousarray(src_pos, dtype=np.int32)
    dst_pos = np.ascontiguousarray(dst_pos, dtype=np.int32)
    n = src_pos.shape[0]
    out_pos = np.zeros((n, 2), dtype=np.int32)
    found = np.zeros(n, dtype=np.uint8)
    ssd = np.zeros(n, dtype=np.int32)
    _chk(lib().mcp_minipatch_find(src._h, dst._h, level, n, src_pos.ctypes.data, dst_pos.ctypes.data, int(rng),
                                  out_pos.ctypes.data, found.ctypes.data, ssd.ctypes.data), "minipatch_find")
    return out_pos, found.astype(bool), ssd


def pack_points(points, handle_of):
    """points: list of dicts(world_pos, pixel_right_w, pixel_down_w, source_kf, source_level, center, fixed)."""
    arr = (TdIn * len(points))()
    for i, p in enumerate(points):
        for k in range(3):
            arr[i].world_pos[k] = p["world_pos"][k]
            arr[i].pixel_right_w[k] = p["pixel_right_w"][k]
            arr[i].pixel_down_w[k] = p["pixel_down_w"][k]
        arr[i].source_kf = handle_of(p["source_kf"])
        arr[i].source_level = int(p["source_level"])
        arr[i].center_x, arr[i].center_y = int(p["center"][0]), int(p["center"][1])
        arr[i].fixed = int(p.get("fixed", 0))
    return arr


def _pose12(R, t):
    return np.ascontiguousarray(np.concatenate([np.asarray(R, dtype=np.float64).reshape(9), np.asarray(t, dtype=np.float64).reshape(3)]))


def track_search(target, cam, base_from_world, cam_from_base, points, rng, subpix_its, exhaustive=False):
    """`points`: list of point dicts, or the ctypes array `pack_points` made from one (map points do not change from
    frame to frame: a caller packs them once)."""
    arr = points if isinstance(points, ctypes.Array) else pack_points(points, lambda kf: kf._h)
    out = np.zeros(len(points), dtype=TD_OUT_DTYPE)
    cs = cam.to_struct()
    b, c = _pose12(*base_from_world), _pose12(*cam_from_base)
    _chk(lib().mcp_track_search(target._h, ctypes.byref(cs), _dp(b), _dp(c), len(points), ctypes.cast(arr, ctypes.c_void_p),
                                int(rng), int(subpix_its), int(exhaustive), out.ctypes.data), "track_search")
    return out


PF_TRACK, PF_REFIND, PF_EPI_COARSE, PF_EPI_REFINE = 0, 1, 2, 3
PF_STATE_DTYPE = np.dtype([("valid", "i4"), ("point_key", "i4"), ("last_warp", "f8", 4), ("template_bad", "i4"), ("jacs_valid", "i4"),
                           ("mean_diff", "f8"), ("templ", "u1", 64), ("jac_templ", "u1", 64)], align=True)
assert PF_STATE_DTYPE.itemsize == 184


class PfTarget(ctypes.Structure):
    _fields_ = [("kf", ctypes.c_void_p), ("cam", ctypes.c_void_p), ("base_from_world", ctypes.c_double * 12), ("cam_from_base", ctypes.c_double * 12)]


class PfItem(ctypes.Structure):
    _fields_ = [("point", TdIn), ("point_key", ctypes.c_int), ("target", ctypes.c_int), ("start_pos", ctypes.c_double * 2)]


def new_pf_states(n):
    """n PatchFinder objects that have seen nothing (include/mcp_img.h mcp_pf_state)."""
    return np.zeros(n, dtype=PF_STATE_DTYPE)


def marshal_patch_sequences(targets, sequences, handle_of, point_handle_of):
    """The argument block of mcp_patch_sequences (include/mcp_img.h) as ctypes objects: target table, sequence offsets, items.
    `handle_of(keyframe)` gives the keyframe handle of a target, `point_handle_of(keyframe)` the source-keyframe handle stored in a
    point record.  Returns (keep-alive tuple, n_targets, table pointer, seq_start array, items pointer, number of items)."""
    cams = [t[1].to_struct() for t in targets]
    tab = (PfTarget * len(targets))()
    for i, (kf, _cam, bfw, cfb) in enumerate(targets):
        tab[i].kf = handle_of(kf)
        tab[i].cam = ctypes.addressof(cams[i])
        b, c = _pose12(*bfw), _pose12(*cfb)
        for k in range(12):
            tab[i].base_from_world[k] = b[k]
            tab[i].cam_from_base[k] = c[k]
    flat = [it for seq in sequences for it in seq]
    seq_start = np.zeros(len(sequences) + 1, dtype=np.int32)
    seq_start[1:] = np.cumsum([len(s_) for s_ in sequences])
    pts = pack_points([it["point"] for it in flat], point_handle_of)
    items = (PfItem * max(len(flat), 1))()
    for i, it in enumerate(flat):
        ctypes.memmove(ctypes.addressof(items[i].point), ctypes.addressof(pts[i]), ctypes.sizeof(TdIn))
        items[i].point_key = int(it.get("point_key", 0))
        items[i].target = int(it.get("target", 0))
        sp = it.get("start_pos", (0.0, 0.0))
        items[i].start_pos[0], items[i].start_pos[1] = float(sp[0]), float(sp[1])
    return (cams, tab, pts, items, seq_start), len(targets), ctypes.cast(tab, ctypes.c_void_p), seq_start, ctypes.cast(items, ctypes.c_void_p), len(flat)


def patch_sequences(mode, targets, sequences, states, rng, subpix_its=0, exhaustive=False):
    """mcp_patch_sequences: `targets` = list of (keyframe, camera, base_from_world (R, t), cam_from_base (R, t)); `sequences` = list of
    lists of items dict(point=<point dict>, point_key, target, start_pos=(x, y)); `states` (new_pf_states) is updated in place.
    Returns one TD_OUT_DTYPE array over all items in order."""
    keep, ntar, tab, seq_start, items, nflat = marshal_patch_sequences(targets, sequences, lambda kf: kf._h, lambda kf: kf._h)
    out = np.zeros(max(nflat, 1), dtype=TD_OUT_DTYPE)
    assert states.dtype == PF_STATE_DTYPE and len(states) == len(sequences)
    rc = lib().mcp_patch_sequences(int(mode), ntar, tab, len(sequences), seq_start.ctypes.data, items, states.ctypes.data, int(rng), int(subpix_its),
                                   int(exhaustive), out.ctypes.data)
    _chk(rc, "patch_sequences")
    del keep
    return out[:nflat]


def track_pose_update(found, found_pos, image_pos, sqrt_inv_noise, jacobian, override_sigma=-1.0, estimator="Tukey"):
    found = np.ascontiguousarray(found, dtype=np.uint8)
    n = found.shape[0]
    fp = np.ascontiguousarray(found_pos, dtype=np.float64)
    ip = np.ascontiguousarray(image_pos, dtype=np.float64)
    si = np.ascontiguousarray(sqrt_inv_noise, dtype=np.float64)
    J = np.ascontiguousarray(jacobian, dtype=np.float64)
    mu = np.zeros(6)
    w = np.zeros(max(n, 1))
    s = ctypes.c_double(0)
    L = lib()
    L.mcp_track_pose_update_m.argtypes = list(L.mcp_track_pose_update.argtypes) + [ctypes.c_int]
    _chk(L.mcp_track_pose_update_m(n, found.ctypes.data, _dp(fp), _dp(ip), _dp(si), _dp(J), float(override_sigma), _dp(mu), _dp(w), ctypes.byref(s), MEST[estimator]), "track_pose_update")
    return mu, w[:n], s.value


def sbi_score(cur, cands):
    """Relocaliser::ScoreKFs (src/Relocaliser.cc:93-121): (index of the first smallest ZMSSD or -1, scores)."""
    n = len(cands)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[None if c is None else c._h for c in cands])
    sc = np.zeros(max(n, 1))
    best = ctypes.c_int(-1)
    _chk(lib().mcp_sbi_score(cur._h, n, ctypes.cast(ptrs, ctypes.c_void_p), sc.ctypes.data, ctypes.byref(best)), "sbi_score")
    return best.value, sc[:n]


def sbi_iterate(cur, target, iterations=6):
    """SmallBlurryImage::IteratePosRelToTarget: (R 2x2, t 2, final score)."""
    se2 = np.zeros(6)
    sc = np.zeros(1)
    _chk(lib().mcp_sbi_iterate(cur._h, target._h, int(iterations), se2.ctypes.data, sc.ctypes.data), "sbi_iterate")
    return se2[:4].reshape(2, 2).copy(), se2[4:].copy(), float(sc[0])


def sbi_se3_from_se2(R2, t2, cam_src, cam_target):
    """SmallBlurryImage::SE3fromSE2 with the 40x30 camera instances: 3x3 rotation."""
    se2 = np.concatenate([np.asarray(R2, dtype=np.float64).ravel(), np.asarray(t2, dtype=np.float64)])
    a, b = cam_src.to_struct(), cam_target.to_struct()
    R = np.zeros(9)
    _chk(lib().mcp_sbi_se3_from_se2(se2.ctypes.data, ctypes.byref(a), ctypes.byref(b), R.ctypes.data), "sbi_se3_from_se2")
    return R.reshape(3, 3)


POSE_POINT_DTYPE = np.dtype([("world_pos", "f8", 3), ("found_pos", "f8", 2), ("sqrt_inv_noise", "f8"), ("image", "f8", 2),
                             ("cam_derivs", "f8", 4), ("cam", "i4"), ("found", "i4")], align=True)

# Tracker::TrackMap's fine-stage schedule (src/Tracker.cc:1063-1075, 800-803): full re-projection at iterations 0, 4 and 9,
# linear updates in between, the 16.0 override only beyond iteration 5
FINE_NONLINEAR = np.array([1, 0, 0, 0, 1, 0, 0, 0, 0, 1], dtype=np.uint8)
FINE_OVERRIDE = np.array([0, 0, 0, 0, 0, 0, 16.0, 16.0, 16.0, 16.0])


def pose_points(world_pos, td_out, cam_index):
    """The mcp_pose_point records of one camera from its map points' world positions and a track_search result."""
    n = len(td_out)
    p = np.zeros(n, dtype=POSE_POINT_DTYPE)
    p["world_pos"] = world_pos
    for f in ("found_pos", "sqrt_inv_noise", "image", "cam_derivs", "found"):
        p[f] = td_out[f]
    p["cam"] = cam_index
    return p


def pose_points_frame(world_pos_per_cam, td_outs):
    """pose_points of all cameras of a frame in one array (camera index = position in the lists)."""
    n = [len(o) for o in td_outs]
    p = np.zeros(sum(n), dtype=POSE_POINT_DTYPE)
    o0 = 0
    for c, (wp, o) in enumerate(zip(world_pos_per_cam, td_outs)):
        q = p[o0:o0 + n[c]]
        q["world_pos"] = wp
        for f in ("found_pos", "sqrt_inv_noise", "image", "cam_derivs", "found"):
            q[f] = o[f]
        q["cam"] = c
        o0 += n[c]
    return p


MEST = {"Tukey": 0, "Cauchy": 1, "Huber": 2}      # Tracker::sMEstimatorName (src/Tracker.cc:1388-1401)


def _refine(fn, pts, cams, cam_from_base, base_from_world, nonlinear, override_sigma, structs, extra=()):
    pts = np.ascontiguousarray(pts, dtype=POSE_POINT_DTYPE).copy()
    n, ncam, nit = len(pts), len(cams), len(nonlinear)
    carr = cams if isinstance(cams, ctypes.Array) else (structs * ncam)(*[c.to_struct() for c in cams])      # (a caller with fixed cameras marshals them once: taylor_camera.camera_array)
    cfb = cam_from_base if isinstance(cam_from_base, np.ndarray) and cam_from_base.ndim == 2 else np.ascontiguousarray(np.stack([_pose12(*T) for T in cam_from_base]))
    bfw = _pose12(*base_from_world).copy()
    nl = np.ascontiguousarray(nonlinear, dtype=np.uint8)
    ov = np.ascontiguousarray(override_sigma, dtype=np.float64)
    mu = np.zeros(6)
    w = np.zeros(max(n, 1))
    rc = fn(n, pts.ctypes.data, ncam, ctypes.cast(carr, ctypes.c_void_p), cfb.ctypes.data, bfw.ctypes.data, nit, nl.ctypes.data, ov.ctypes.data,
            mu.ctypes.data, w.ctypes.data, *extra)
    return rc, (bfw[:9].reshape(3, 3).copy(), bfw[9:].copy()), mu, w[:n], pts


def track_pose_refine_sharded(pts, cams, cam_from_base, base_from_world, allreduce=None, rank=0, world=1, cap=None,
                              nonlinear=FINE_NONLINEAR, override_sigma=FINE_OVERRIDE):
    """The pose iterations with the cameras spread over ranks (one camera per GPU, BASELINE config c5): `pts` are THIS rank's points,
    `allreduce(device_ptr, count, stream)` an in-place SUM all-reduce of doubles (e.g. mcptam_amd.dist.RcclAllReduce / GlooAllReduce);
    `cap` >= the largest point count of any rank.  Returns what track_pose_refine returns; the pose is the same on every rank."""
    def tramp(_user, buf, count, stream):
        try:
            allreduce(int(buf), int(count), int(stream or 0))
            return 0
        except Exception as exc:      # never let an exception cross the C ABI
            print("all-reduce hook raised:", repr(exc), flush=True)
            return 1
    hook = _cb.ALLREDUCE_FN(tramp) if allreduce is not None else ctypes.cast(None, _cb.ALLREDUCE_FN)
    cap = max(len(pts), 1) if cap is None else int(cap)

    def fn(n, p, ncam, carr, cfb, bfw, nit, nl, ov, mu, w):
        return lib().mcp_track_pose_refine_sharded(n, p, ncam, carr, cfb, bfw, nit, nl, ov, mu, w, hook, None, int(rank), int(world), cap)
    rc, pose, mu, w, out = _refine(fn, pts, cams, cam_from_base, base_from_world, nonlinear, override_sigma, McpCamera)
    _chk(rc, "track_pose_refine_sharded")
    return pose, mu, w, out


class TrackFrame:
    """mcp_track_frame with everything that does not change from frame to frame marshalled once (what a native caller keeps in its own
    structs): the keyframes of the rig, the camera models, CamFromBase, the packed points of every camera, and -- `stateful=True` -- the
    persistent PatchFinder members of every (point, camera) (mcp_pf_state, point keys = position in the packed list unless given).
    `run(imgs, base_from_world, ...)` = one stage of Tracker::TrackMap: MakeKeyFrame_Lite of every camera (imgs=None: skip), the search,
    the pose iterations.  Returns (td_outs per camera, pose points, (R, t), mu, weights)."""

    def __init__(self, kfs, cams, cams_from_base, points, stateful=False, point_keys=None, pinned=True):
        from .taylor_camera import camera_array
        self.kfs = list(kfs)
        n = self.n = len(self.kfs)
        self.hs = (ctypes.c_void_p * n)(*[k._h for k in self.kfs])
        self.cs = cams if isinstance(cams, ctypes.Array) else camera_array(cams)
        self.cfb = np.ascontiguousarray(cams_from_base.reshape(-1)) if isinstance(cams_from_base, np.ndarray) else np.ascontiguousarray(np.concatenate([_pose12(*c) for c in cams_from_base]))
        self.arrs = [p if isinstance(p, ctypes.Array) else pack_points(p, lambda kf: kf._h) for p in points]
        self.lens = [len(a) for a in self.arrs]
        self.total = sum(self.lens)
        self.offs = np.concatenate([[0], np.cumsum(self.lens)]).astype(int)
        self.ns = (ctypes.c_int * n)(*self.lens)
        self.ins = (ctypes.c_void_p * n)(*[ctypes.cast(a, ctypes.c_void_p) for a in self.arrs])
        # result buffers in pinned host memory (a native tracker registers its TrackerData arrays the same way): the copies back are DMA
        from . import hip_rt
        self.whole, self._own0 = hip_rt.pinned_zeros(max(self.total, 1), TD_OUT_DTYPE) if pinned else (np.zeros(max(self.total, 1), dtype=TD_OUT_DTYPE), None)
        self.ops = (ctypes.c_void_p * n)(*[self.whole.ctypes.data + int(self.offs[c])*TD_OUT_DTYPE.itemsize for c in range(n)])
        self.pts, self._own1 = hip_rt.pinned_zeros(max(self.total, 1), POSE_POINT_DTYPE) if pinned else (np.zeros(max(self.total, 1), dtype=POSE_POINT_DTYPE), None)
        self.w, self._own2 = hip_rt.pinned_zeros(max(self.total, 1), np.float64) if pinned else (np.zeros(max(self.total, 1)), None)
        self.mu = np.zeros(6)
        self.states = self.keys = self.sp = self.kp = None
        if stateful:
            self.states = [new_pf_states(m) for m in self.lens]
            self.keys = [np.ascontiguousarray(point_keys[c] if point_keys is not None else np.arange(self.lens[c]), dtype=np.int32) for c in range(n)]
            self.sp = (ctypes.c_void_p * n)(*[s_.ctypes.data for s_ in self.states])
            self.kp = (ctypes.c_void_p * n)(*[k_.ctypes.data for k_ in self.keys])

    def run(self, imgs, base_from_world, rng, subpix_its, exhaustive=False, nonlinear=FINE_NONLINEAR, override_sigma=FINE_OVERRIDE, estimator="Tukey",
            on_device=False, strides=None, want_points=True, view=False):
        """view=True: the TrackerData results are not copied into this object's arrays; the returned per-camera arrays are views of the
        library's pinned block (mcp_track_frame_view), valid until the next frame."""
        n = self.n
        ip = st = None
        keep = None
        if imgs is not None:
            if on_device:
                ip = (ctypes.c_void_p * n)(*[int(a) for a in imgs])
                st = (ctypes.c_int * n)(*[int(s_) for s_ in (strides or [k.w for k in self.kfs])])
            else:
                keep = [np.ascontiguousarray(a, dtype=np.uint8) for a in imgs]
                ip = (ctypes.c_void_p * n)(*[a.ctypes.data for a in keep])
                st = (ctypes.c_int * n)(*[a.strides[0] for a in keep])
        bfw = _pose12(*base_from_world).copy()
        nl = np.ascontiguousarray(nonlinear, dtype=np.uint8)
        ov = np.ascontiguousarray(override_sigma, dtype=np.float64)
        import time as _time
        fn = lib().mcp_track_frame
        t0 = _time.perf_counter()
        rc = fn(n, self.hs, ip, st, int(on_device), None, ctypes.cast(self.cs, ctypes.c_void_p), bfw.ctypes.data, self.cfb.ctypes.data, self.ns, self.ins,
                self.kp, self.sp, int(rng), int(subpix_its), int(exhaustive), len(nl), nl.ctypes.data, ov.ctypes.data, MEST[estimator],
                None if view else self.ops, self.pts.ctypes.data if want_points else None, self.mu.ctypes.data, self.w.ctypes.data)
        self.abi_seconds = _time.perf_counter() - t0          # the C call alone: what a native caller pays for the frame
        _chk(rc, "track_frame")
        if view:
            L = lib()
            L.mcp_track_frame_view.restype = ctypes.c_void_p
            L.mcp_track_frame_view.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
            outs = []
            for c in range(n):
                cnt = ctypes.c_int(0)
                ptr = L.mcp_track_frame_view(self.kfs[0]._h, c, ctypes.byref(cnt))
                if cnt.value != self.lens[c]:
                    raise RuntimeError("mcp_track_frame_view: " + _cb.last_error())
                outs.append(np.frombuffer((ctypes.c_char * (cnt.value * TD_OUT_DTYPE.itemsize)).from_address(ptr), dtype=TD_OUT_DTYPE) if cnt.value else np.zeros(0, dtype=TD_OUT_DTYPE))
        else:
            outs = [self.whole[self.offs[c]:self.offs[c + 1]] for c in range(n)]
        return outs, (self.pts[:self.total] if want_points else None), (bfw[:9].reshape(3, 3).copy(), bfw[9:].copy()), self.mu.copy(), self.w[:self.total]


def track_pose_refine(pts, cams, cam_from_base, base_from_world, nonlinear=FINE_NONLINEAR, override_sigma=FINE_OVERRIDE, estimator="Tukey"):
    """All Gauss-Newton pose iterations of one frame in one device launch.  Returns (BaseFromWorld (R, t), last update,
    last M-estimator weights, points with their final image positions)."""
    L = lib()
    L.mcp_track_pose_refine_m.argtypes = list(L.mcp_track_pose_refine.argtypes) + [ctypes.c_int]
    rc, pose, mu, w, out = _refine(L.mcp_track_pose_refine_m, pts, cams, cam_from_base, base_from_world, nonlinear, override_sigma, McpCamera, extra=(MEST[estimator],))
    _chk(rc, "track_pose_refine")
    return pose, mu, w, out
