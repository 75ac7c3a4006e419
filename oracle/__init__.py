"""ctypes loader for the CPU oracle (liborc.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from mcptam_amd/ (the product path).  PARITY UNPINNED: see
oracle/ba_oracle.h.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)


class OrcIterLog(ctypes.Structure):
    _fields_ = [("chi2_start", ctypes.c_double), ("chi2_end", ctypes.c_double),
                ("lambda_end", ctypes.c_double), ("sigma_sq", ctypes.c_double),
                ("rms_update", ctypes.c_double), ("trials", ctypes.c_int), ("accepted", ctypes.c_int)]


def build(force=False):
    if os.environ.get("ORC_LIB"):          # bench.py: the -O3 -march=native -fopenmp build made on the measuring box (`make native`)
        return os.environ["ORC_LIB"]
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("ba_oracle.c", "ba_oracle.h", "img_oracle.c", "img_oracle.h", "ba_baseline.inc", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.orc_ba_create.restype = ctypes.c_void_p
        L.orc_ba_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_ba_destroy.argtypes = [ctypes.c_void_p]
        L.orc_ba_set_limits.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        L.orc_ba_disable_convergence.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_ba_set_solver.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.orc_ba_threads.argtypes = [ctypes.c_void_p]
        L.orc_ba_add_pose.argtypes = [ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_int]
        L.orc_ba_add_point.argtypes = [ctypes.c_void_p, c_double_p, c_int_p, ctypes.c_int, ctypes.c_int]
        L.orc_ba_add_meas.argtypes = [ctypes.c_void_p, c_int_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_double, ctypes.c_int]
        L.orc_ba_compute.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ubyte), ctypes.c_int, ctypes.c_double]
        for f in ("orc_ba_converged", "orc_ba_total_iterations", "orc_ba_num_outliers", "orc_ba_num_iter_logs",
                  "orc_ba_num_meas", "orc_ba_prepare"):
            getattr(L, f).argtypes = [ctypes.c_void_p]
        for f in ("orc_ba_sigma_squared", "orc_ba_mean_chi_squared", "orc_ba_max_cov", "orc_ba_lambda"):
            getattr(L, f).argtypes = [ctypes.c_void_p]
            getattr(L, f).restype = ctypes.c_double
        L.orc_ba_get_point.argtypes = [ctypes.c_void_p, ctypes.c_int, c_double_p]
        L.orc_ba_get_pose.argtypes = [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p]
        L.orc_ba_get_outliers.argtypes = [ctypes.c_void_p, c_int_p, ctypes.c_int]
        L.orc_ba_get_iter_logs.argtypes = [ctypes.c_void_p, ctypes.POINTER(OrcIterLog), ctypes.c_int]
        L.orc_ba_eval.argtypes = [ctypes.c_void_p, c_double_p, c_double_p]
        L.orc_ba_jacobian.argtypes = [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p, c_double_p]
        L.orc_ba_numeric_jacobian.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, c_double_p, c_double_p, c_double_p]
        L.orc_ba_debug_solve.argtypes = [ctypes.c_void_p, ctypes.c_double, c_double_p, c_double_p]
        L.orc_ba_debug_system.argtypes = [ctypes.c_void_p, ctypes.c_double, c_double_p]
        L.orc_ba_debug_robust_chi2.argtypes = [ctypes.c_void_p, c_double_p]
        L.orc_ba_debug_robust_chi2.restype = ctypes.c_double
        L.orc_cam_project.argtypes = [ctypes.c_void_p, c_double_p, c_double_p, c_double_p]
        L.orc_cam_sphere_deriv.argtypes = [c_double_p, c_double_p, c_double_p]
        L.orc_se3_exp.argtypes = [c_double_p, c_double_p, c_double_p]
        L.orc_so3_exp.argtypes = [c_double_p, c_double_p]
        L.orc_huber_sigma_squared.argtypes = [c_double_p, ctypes.c_int]
        L.orc_huber_sigma_squared.restype = ctypes.c_double
        L.orc_tukey_sigma_squared.argtypes = [c_double_p, ctypes.c_int]
        L.orc_tukey_sigma_squared.restype = ctypes.c_double
        L.orc_tukey_weight.argtypes = [ctypes.c_double, ctypes.c_double]
        L.orc_tukey_weight.restype = ctypes.c_double
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def cam_project(cam, xc):
    """orc_cam_project: (uv, D (2x2 projection derivatives, row-major), invalid flag) of a camera-frame point."""
    import numpy as np
    st = cam.to_struct()
    xc = np.ascontiguousarray(xc, dtype=np.float64)
    uv = np.zeros(2)
    D = np.zeros(4)
    inv = lib().orc_cam_project(ctypes.byref(st), _dp(xc), _dp(uv), _dp(D))
    return uv, D.reshape(2, 2), bool(inv)


class OracleBundle:
    """ChainBundle-shaped wrapper over the oracle (same surface as mcptam_amd.ChainBundle)."""

    def __init__(self, cams, use_robust=True, use_tukey=True, verbose=False):
        from mcptam_amd.taylor_camera import camera_array
        self._L = lib()
        self._cams = camera_array(cams)
        self._h = self._L.orc_ba_create(ctypes.cast(self._cams, ctypes.c_void_p), len(cams), int(use_robust), int(use_tukey), int(verbose))
        self.abort = ctypes.c_ubyte(0)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_ba_destroy(self._h)
            self._h = None

    def SetLimits(self, max_trials=100, pct_limit=1e-10, rms_limit=1e-10, min_sigma=0.5):
        self._L.orc_ba_set_limits(self._h, max_trials, pct_limit, rms_limit, min_sigma)

    def SetSolver(self, solver, threads=1):
        """CPU-baseline variants (oracle/ba_baseline.inc): 0 = the oracle proper, 1 = A sparse L D L^T of the un-marginalised
        system (1 thread), 2 = B Schur + OpenMP.  Returns the thread count in effect (1 without OpenMP in the loaded build)."""
        self._L.orc_ba_set_solver(self._h, int(solver), int(threads))
        return self._L.orc_ba_threads(self._h)

    def SetDupSymmetric(self, on=True):
        """oracle-only switch: symmetric cross terms for a pose vertex that occurs twice in an edge (ba_oracle.c build_system)"""
        self._L.orc_ba_set_dup_symmetric.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self._L.orc_ba_set_dup_symmetric(self._h, int(on))

    def SetVariant(self, key, value):
        """oracle-only [3P-memory] switches of the LM schedule (ba_oracle.h orc_ba_set_variant)"""
        self._L.orc_ba_set_variant.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
        self._L.orc_ba_set_variant(self._h, int(key), float(value))

    def SetFailTrial(self, k):
        """test switch: the k-th LM trial behaves as a failed factorisation"""
        self._L.orc_ba_set_fail_trial.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self._L.orc_ba_set_fail_trial(self._h, int(k))

    def DisableConvergence(self, disable=True):
        self._L.orc_ba_disable_convergence(self._h, int(disable))

    def AddPose(self, R, t, fixed):
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
        t = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
        return self._L.orc_ba_add_pose(self._h, _dp(R), _dp(t), int(fixed))

    def AddPoint(self, x, chain, fixed):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(3)
        c = np.ascontiguousarray(chain, dtype=np.int32)
        r = self._L.orc_ba_add_point(self._h, _dp(x), _ip(c), len(c), int(fixed))
        if r < 0:
            raise ValueError("AddPoint: bad chain")
        return r

    def AddMeas(self, chain, point_id, uv, sigma_sq, cam_index):
        c = np.ascontiguousarray(chain, dtype=np.int32)
        uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(2)
        if self._L.orc_ba_add_meas(self._h, _ip(c), len(c), int(point_id), _dp(uv), float(sigma_sq), int(cam_index)) < 0:
            raise ValueError("AddMeas: bad arguments")

    def Compute(self, n_iter=100, user_lambda=-1.0):
        return self._L.orc_ba_compute(self._h, ctypes.byref(self.abort), int(n_iter), float(user_lambda))

    def Converged(self):
        return bool(self._L.orc_ba_converged(self._h))

    def TotalIterations(self):
        return self._L.orc_ba_total_iterations(self._h)

    def GetPoint(self, pid):
        x = np.zeros(3)
        assert self._L.orc_ba_get_point(self._h, int(pid), _dp(x)) == 0
        return x

    def GetPose(self, pid):
        R = np.zeros(9)
        t = np.zeros(3)
        assert self._L.orc_ba_get_pose(self._h, int(pid), _dp(R), _dp(t)) == 0
        return R.reshape(3, 3), t

    def GetOutlierMeasurements(self):
        n = self._L.orc_ba_num_outliers(self._h)
        out = np.zeros((max(n, 1), 3), dtype=np.int32)
        n = self._L.orc_ba_get_outliers(self._h, _ip(out), n)
        return [tuple(int(v) for v in out[i]) for i in range(n)]

    def GetSigmaSquared(self):
        return self._L.orc_ba_sigma_squared(self._h)

    def GetMeanChiSquared(self):
        return self._L.orc_ba_mean_chi_squared(self._h)

    def GetMaxCov(self):
        return self._L.orc_ba_max_cov(self._h)

    def GetLambda(self):
        return self._L.orc_ba_lambda(self._h)

    def IterLogs(self):
        n = self._L.orc_ba_num_iter_logs(self._h)
        arr = (OrcIterLog * max(n, 1))()
        n = self._L.orc_ba_get_iter_logs(self._h, arr, n)
        return [dict(chi2_start=a.chi2_start, chi2_end=a.chi2_end, lambda_end=a.lambda_end, sigma_sq=a.sigma_sq,
                     rms_update=a.rms_update, trials=a.trials, accepted=a.accepted) for a in arr[:n]]

    # ---- introspection for the self-consistency tests ----
    def Prepare(self):
        return self._L.orc_ba_prepare(self._h)

    def NumMeas(self):
        return self._L.orc_ba_num_meas(self._h)

    def Eval(self):
        m = self.NumMeas()
        chi2 = np.zeros(m)
        err = np.zeros((m, 2))
        self._L.orc_ba_eval(self._h, _dp(chi2), _dp(err))
        return chi2, err

    MAX_CHAIN = 8        # ORC_MAX_CHAIN: the Jacobian mask has observer links in bits [0, 8), source links in [8, 16), the point in bit 16

    def Jacobian(self, m, numeric=False, delta=1e-6):
        jo = np.zeros((self.MAX_CHAIN, 2, 6))
        js = np.zeros((self.MAX_CHAIN, 2, 6))
        jp = np.zeros((2, 3))
        if numeric:
            mask = self._L.orc_ba_numeric_jacobian(self._h, int(m), float(delta), _dp(jo), _dp(js), _dp(jp))
        else:
            mask = self._L.orc_ba_jacobian(self._h, int(m), _dp(jo), _dp(js), _dp(jp))
        return mask, jo, js, jp

    def DebugSolve(self, lam):
        n = self.Prepare()
        xs = np.zeros(n)
        xd = np.zeros(n)
        rc = self._L.orc_ba_debug_solve(self._h, float(lam), _dp(xs), _dp(xd))
        return rc, xs, xd

    def DebugSystem(self, lam):
        """(S, rhs, pose part of J^T r) of the reduced pose system at the current state."""
        n = self._L.orc_ba_debug_system(self._h, float(lam), None)
        out = np.zeros(n * n + 2 * n)
        if self._L.orc_ba_debug_system(self._h, float(lam), _dp(out)) < 0:
            raise RuntimeError("orc_ba_debug_system: a point block is not positive definite")
        return out[:n * n].reshape(n, n), out[n * n:n * n + n], out[n * n + n:]

    def DebugRobustChi2(self):
        s = ctypes.c_double(0)
        c = self._L.orc_ba_debug_robust_chi2(self._h, ctypes.byref(s))
        return c, s.value


# ------------------------------------------------------------------ image path oracle
class OrcTdIn(ctypes.Structure):
    _fields_ = [("world_pos", ctypes.c_double * 3), ("pixel_right_w", ctypes.c_double * 3), ("pixel_down_w", ctypes.c_double * 3),
                ("source_kf", ctypes.c_void_p), ("source_level", ctypes.c_int), ("center_x", ctypes.c_int),
                ("center_y", ctypes.c_int), ("fixed", ctypes.c_int)]


_IMG_BOUND = False


def img_lib():
    global _IMG_BOUND
    L = lib()
    if not _IMG_BOUND:
        L.orc_kf_create.restype = ctypes.c_void_p
        L.orc_kf_create.argtypes = [ctypes.c_int] * 5
        L.orc_kf_destroy.argtypes = [ctypes.c_void_p]
        L.orc_kf_make_lite.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_kf_level_size.argtypes = [ctypes.c_void_p, ctypes.c_int, c_int_p, c_int_p]
        L.orc_kf_image.restype = ctypes.c_void_p
        L.orc_kf_image.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_kf_num_corners.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_kf_num_prev.argtypes = [ctypes.c_void_p]
        L.orc_kf_corners.restype = ctypes.c_void_p
        L.orc_kf_corners.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_kf_row_lut.restype = ctypes.c_void_p
        L.orc_kf_row_lut.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_kf_fast_thresh.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_kf_fast_frequency.restype = ctypes.c_void_p
        L.orc_kf_fast_frequency.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_kf_make_rest.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]
        L.orc_kf_num_candidates.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_kf_get_candidates.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, c_double_p, ctypes.c_int]
        L.orc_fast10_is_corner.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.orc_fast10_score.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.orc_shi_tomasi.restype = ctypes.c_double
        L.orc_shi_tomasi.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_minipatch_find.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_track_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.orc_track_pose_update.argtypes = [ctypes.c_int, ctypes.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                            ctypes.c_double, c_double_p, c_double_p, c_double_p]
        L.orc_kf_make_sbi.argtypes = [ctypes.c_void_p, ctypes.c_double]
        L.orc_track_pose_refine.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_track_pose_refine_m.argtypes = list(L.orc_track_pose_refine.argtypes) + [ctypes.c_int]
        for f in ("orc_kf_sbi_small", "orc_kf_sbi_template", "orc_kf_sbi_jacs"):
            getattr(L, f).restype = ctypes.c_void_p
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.orc_sbi_zmssd.restype = ctypes.c_double
        L.orc_sbi_zmssd.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_sbi_score.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, c_double_p]
        L.orc_sbi_iterate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p]
        L.orc_sbi_se3_from_se2.argtypes = [c_double_p, ctypes.c_void_p, ctypes.c_void_p, c_double_p]
        _IMG_BOUND = True
    return L


class OracleKeyFrame:
    def __init__(self, w, h, adaptive=True, glare=False, pavgb=False):
        self._L = img_lib()
        self._h = self._L.orc_kf_create(int(w), int(h), int(adaptive), int(glare), int(pavgb))
        self.w, self.h = w, h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_kf_destroy(self._h)
            self._h = None

    def MakeKeyFrame_Lite(self, img, masks=None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        mp = None
        if masks is not None:
            self._masks = [None if m is None else np.ascontiguousarray(m, dtype=np.uint8) for m in masks]
            arr = (ctypes.c_void_p * 4)(*[None if m is None else m.ctypes.data for m in self._masks])
            mp = ctypes.cast(arr, ctypes.c_void_p)
        self._L.orc_kf_make_lite(self._h, img.ctypes.data, img.strides[0], mp)

    def NumPrev(self):
        return int(self._L.orc_kf_num_prev(self._h))

    def LevelSize(self, level):
        w, h = ctypes.c_int(), ctypes.c_int()
        self._L.orc_kf_level_size(self._h, level, ctypes.byref(w), ctypes.byref(h))
        return w.value, h.value

    def Image(self, level):
        w, h = self.LevelSize(level)
        p = self._L.orc_kf_image(self._h, level)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_ubyte)), shape=(h, w)).copy()

    def Corners(self, level):
        n = self._L.orc_kf_num_corners(self._h, level)
        if n == 0:
            return np.zeros((0, 2), dtype=np.int32)
        p = self._L.orc_kf_corners(self._h, level)
        return np.ctypeslib.as_array(ctypes.cast(p, c_int_p), shape=(n, 2)).copy()

    def RowLUT(self, level):
        _, h = self.LevelSize(level)
        p = self._L.orc_kf_row_lut(self._h, level)
        return np.ctypeslib.as_array(ctypes.cast(p, c_int_p), shape=(h,)).copy()

    def FastThresh(self, level):
        return self._L.orc_kf_fast_thresh(self._h, level)

    def FastFrequency(self, level):
        p = self._L.orc_kf_fast_frequency(self._h, level)
        return np.ctypeslib.as_array(ctypes.cast(p, c_double_p), shape=(31,)).copy()

    def MakeKeyFrame_Rest(self, use_shi=False, use_percent=True, top_fraction=0.8, thresh=70.0, nonmax_score=0):
        self._L.orc_kf_make_rest(self._h, int(use_shi), int(use_percent), float(top_fraction), float(thresh), int(nonmax_score))

    def Candidates(self, level):
        n = self._L.orc_kf_num_candidates(self._h, level)
        pos = np.zeros((max(n, 1), 2), dtype=np.int32)
        sc = np.zeros(max(n, 1))
        n = self._L.orc_kf_get_candidates(self._h, level, pos.ctypes.data, _dp(sc), n)
        return pos[:n], sc[:n]


    # ---- SmallBlurryImage (src/SmallBlurryImage.cc)
    def MakeSBI(self, blur=2.5):
        self._L.orc_kf_make_sbi(self._h, float(blur))

    def SBI(self):
        """(mimSmall u8 30x40, mimTemplate f32 30x40, mimImageJacs f32 30x40x2)"""
        def arr(fn, ct, n):
            return np.ctypeslib.as_array(ctypes.cast(fn(self._h), ctypes.POINTER(ct)), shape=(n,)).copy()
        return (arr(self._L.orc_kf_sbi_small, ctypes.c_uint8, 1200).reshape(30, 40),
                arr(self._L.orc_kf_sbi_template, ctypes.c_float, 1200).reshape(30, 40),
                arr(self._L.orc_kf_sbi_jacs, ctypes.c_float, 2400).reshape(30, 40, 2))


def oracle_sbi_score(cur, cands):
    """Relocaliser::ScoreKFs: (best index or -1, scores)."""
    n = len(cands)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[c._h for c in cands])
    sc = np.zeros(max(n, 1))
    best = img_lib().orc_sbi_score(cur._h, n, ctypes.cast(ptrs, ctypes.c_void_p), _dp(sc))
    return best, sc[:n]


def oracle_sbi_iterate(cur, target, iterations=6):
    """SmallBlurryImage::IteratePosRelToTarget: (R 2x2, t 2, score)."""
    se2 = np.zeros(6)
    sc = np.zeros(1)
    img_lib().orc_sbi_iterate(cur._h, target._h, int(iterations), _dp(se2), _dp(sc))
    return se2[:4].reshape(2, 2).copy(), se2[4:].copy(), float(sc[0])


def oracle_sbi_se3_from_se2(R2, t2, cam_src, cam_target):
    se2 = np.concatenate([np.asarray(R2, dtype=np.float64).ravel(), np.asarray(t2, dtype=np.float64)])
    a, b = cam_src.to_struct(), cam_target.to_struct()
    R = np.zeros(9)
    img_lib().orc_sbi_se3_from_se2(_dp(se2), ctypes.byref(a), ctypes.byref(b), _dp(R))
    return R.reshape(3, 3)


def oracle_minipatch_find(src, dst, level, src_pos, dst_pos, rng):
    src_pos = np.ascontiguousarray(src_pos, dtype=np.int32)
    dst_pos = np.ascontiguousarray(dst_pos, dtype=np.int32)
    n = src_pos.shape[0]
    out_pos = np.zeros((n, 2), dtype=np.int32)
    found = np.zeros(n, dtype=np.uint8)
    ssd = np.zeros(n, dtype=np.int32)
    img_lib().orc_minipatch_find(src._h, dst._h, level, n, src_pos.ctypes.data, dst_pos.ctypes.data, int(rng), out_pos.ctypes.data, found.ctypes.data, ssd.ctypes.data)
    return out_pos, found.astype(bool), ssd


def oracle_track_search(target, cam, base_from_world, cam_from_base, points, rng, subpix_its, exhaustive=False):
    from mcptam_amd.keyframe import TD_OUT_DTYPE, _pose12
    arr = (OrcTdIn * len(points))()
    for i, p in enumerate(points):
        for k in range(3):
            arr[i].world_pos[k] = p["world_pos"][k]
            arr[i].pixel_right_w[k] = p["pixel_right_w"][k]
            arr[i].pixel_down_w[k] = p["pixel_down_w"][k]
        arr[i].source_kf = p["source_kf_oracle"]._h
        arr[i].source_level = int(p["source_level"])
        arr[i].center_x, arr[i].center_y = int(p["center"][0]), int(p["center"][1])
        arr[i].fixed = int(p.get("fixed", 0))
    out = np.zeros(len(points), dtype=TD_OUT_DTYPE)
    cs = cam.to_struct()
    b, c = _pose12(*base_from_world), _pose12(*cam_from_base)
    img_lib().orc_track_search(target._h, ctypes.byref(cs), _dp(b), _dp(c), len(points), ctypes.cast(arr, ctypes.c_void_p), int(rng), int(subpix_its), int(exhaustive), out.ctypes.data)
    return out


def oracle_patch_sequences(mode, targets, sequences, states, rng, subpix_its=0, exhaustive=False):
    """CPU restatement of the stateful PatchFinder flows (same signature as mcptam_amd.keyframe.patch_sequences); the point dicts
    carry the oracle's keyframe under 'source_kf_oracle', targets hold OracleKeyFrame objects."""
    from mcptam_amd import keyframe as kf
    L = img_lib()
    L.orc_patch_sequences.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    seqs = [[dict(it, point=dict(it["point"], source_kf=it["point"]["source_kf_oracle"]._h)) for it in seq] for seq in sequences]
    # the C structs are the C ABI's (include/mcp_img.h), so the product binding's marshaller lays them out; the call is the oracle's own
    keep, ntar, tab, seq_start, items, nflat = kf.marshal_patch_sequences(targets, seqs, lambda k: k._h, lambda h: h)
    out = np.zeros(max(nflat, 1), dtype=kf.TD_OUT_DTYPE)
    assert states.dtype == kf.PF_STATE_DTYPE and len(states) == len(sequences)
    rc = L.orc_patch_sequences(int(mode), ntar, tab, len(sequences), seq_start.ctypes.data, items, states.ctypes.data, int(rng), int(subpix_its),
                               int(exhaustive), out.ctypes.data)
    assert rc == 0
    del keep
    return out[:nflat]


def oracle_track_pose_update(found, found_pos, image_pos, sqrt_inv_noise, jacobian, override_sigma=-1.0, estimator="Tukey"):
    found = np.ascontiguousarray(found, dtype=np.uint8)
    n = found.shape[0]
    fp = np.ascontiguousarray(found_pos, dtype=np.float64)
    ip = np.ascontiguousarray(image_pos, dtype=np.float64)
    si = np.ascontiguousarray(sqrt_inv_noise, dtype=np.float64)
    J = np.ascontiguousarray(jacobian, dtype=np.float64)
    mu = np.zeros(6)
    w = np.zeros(max(n, 1))
    s = ctypes.c_double(0)
    from mcptam_amd.keyframe import MEST
    L = img_lib()
    L.orc_track_pose_update_m.argtypes = list(L.orc_track_pose_update.argtypes) + [ctypes.c_int]
    L.orc_track_pose_update_m(n, found.ctypes.data, _dp(fp), _dp(ip), _dp(si), _dp(J), float(override_sigma), _dp(mu), _dp(w), ctypes.byref(s), MEST[estimator])
    return mu, w[:n], s.value


def oracle_track_pose_refine(pts, cams, cam_from_base, base_from_world, nonlinear=None, override_sigma=None, estimator="Tukey"):
    """CPU restatement of Tracker::TrackMap's pose iterations (same signature as mcptam_amd.keyframe.track_pose_refine)."""
    from mcptam_amd import keyframe as kf
    from mcptam_amd.taylor_camera import McpCamera
    nonlinear = kf.FINE_NONLINEAR if nonlinear is None else nonlinear
    override_sigma = kf.FINE_OVERRIDE if override_sigma is None else override_sigma
    rc, pose, mu, w, out = kf._refine(img_lib().orc_track_pose_refine_m, pts, cams, cam_from_base, base_from_world, nonlinear, override_sigma, McpCamera, extra=(kf.MEST[estimator],))
    return pose, mu, w, out
