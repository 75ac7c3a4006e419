"""ctypes binding of the C-ABI library (include/mcp_ba.h) with the ChainBundle surface.

`ChainBundle` carries the same method names and argument meaning as the reference class
(/root/reference/include/mcptam/ChainBundle.h:106-186) so the parity tests read like calls
to the reference.  Everything numeric happens inside libmcptam_hip.so on the GPU; if the
library or a gfx950 device is missing this module raises -- there is no CPU path here.
"""
import ctypes
import os
import time

import numpy as np

from .taylor_camera import camera_array

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCP_HIP_LIB") or os.path.join(_HERE, "libmcptam_hip.so")      # MCP_HIP_LIB: a build variant (scripts/build_variants.sh)
_LIB = None

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_ubyte_p = ctypes.POINTER(ctypes.c_ubyte)


class McpBaParams(ctypes.Structure):
    _fields_ = [("max_iterations", ctypes.c_int), ("max_trials_after_failure", ctypes.c_int),
                ("update_percent_limit", ctypes.c_double), ("update_rms_limit", ctypes.c_double),
                ("min_mestimator_sigma", ctypes.c_double), ("disable_convergence", ctypes.c_int),
                ("device", ctypes.c_int), ("profile", ctypes.c_int)]


class McpBaIterLog(ctypes.Structure):
    _fields_ = [("chi2_start", ctypes.c_double), ("chi2_end", ctypes.c_double), ("lambda_end", ctypes.c_double),
                ("sigma_sq", ctypes.c_double), ("rms_update", ctypes.c_double), ("trials", ctypes.c_int),
                ("accepted", ctypes.c_int)]


class McpBaTiming(ctypes.Structure):
    _fields_ = [("total_ms", ctypes.c_double), ("structure_ms", ctypes.c_double), ("eval_ms", ctypes.c_double),
                ("select_ms", ctypes.c_double), ("linearize_ms", ctypes.c_double), ("schur_ms", ctypes.c_double),
                ("cholesky_ms", ctypes.c_double), ("solve_ms", ctypes.c_double), ("update_ms", ctypes.c_double),
                ("n_linearize", ctypes.c_int), ("n_trials", ctypes.c_int), ("n_solves", ctypes.c_int),
                ("n_spec_hits", ctypes.c_int),
                ("n_collectives_main", ctypes.c_int), ("n_collectives_spec", ctypes.c_int),
                ("collective_bytes_main", ctypes.c_double), ("collective_bytes_spec", ctypes.c_double),
                ("n_median_fast", ctypes.c_int), ("n_persist_fallbacks", ctypes.c_int),
                ("schur_mfma_per_system", ctypes.c_double), ("schur_flops_structural", ctypes.c_double),
                ("chol_flops_plan", ctypes.c_double), ("chol_chains", ctypes.c_int)]


ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)

# every symbol include/mcp_ba.h declares (checked by the CPU test-suite)
BA_SYMBOLS = [
    "mcp_last_error", "mcp_device_count", "mcp_ba_create", "mcp_ba_destroy", "mcp_ba_add_pose", "mcp_ba_add_point",
    "mcp_ba_add_meas", "mcp_ba_add_points", "mcp_ba_add_measurements", "mcp_ba_compute", "mcp_ba_converged",
    "mcp_ba_total_iterations", "mcp_ba_get_point", "mcp_ba_get_pose", "mcp_ba_get_points", "mcp_ba_get_poses",
    "mcp_ba_num_outliers", "mcp_ba_get_outliers", "mcp_ba_sigma_squared", "mcp_ba_mean_chi_squared", "mcp_ba_max_cov",
    "mcp_ba_lambda", "mcp_ba_num_iter_logs", "mcp_ba_get_iter_logs", "mcp_ba_get_timing", "mcp_ba_set_allreduce",
    "mcp_ba_prepare", "mcp_ba_eval", "mcp_ba_robust_chi2", "mcp_ba_debug_solve", "mcp_dense_spd_solve",
    "mcp_dense_spd_stress", "mcp_ba_debug_system", "mcp_chol_debug_factor", "mcp_chol_time", "mcp_debug_pose_cut",
    "mcp_ba_struct_cache_stats", "mcp_ba_struct_cache_near_hits", "mcp_ba_struct_cache_clear",
    "mcp_comm_unique_id", "mcp_comm_init", "mcp_comm_destroy", "mcp_ba_set_comm", "mcp_comm_allreduce", "mcp_comm_allreduce_lane",
]


def lib():
    """Load libmcptam_hip.so (built by `__graft_entry__.build()` / mcptam_amd/csrc/Makefile)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libmcptam_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "the HIP path has no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    L.mcp_last_error.restype = ctypes.c_char_p
    L.mcp_ba_create.restype = ctypes.c_void_p
    L.mcp_ba_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.mcp_ba_destroy.argtypes = [ctypes.c_void_p]
    L.mcp_ba_add_pose.argtypes = [ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_int]
    L.mcp_ba_add_point.argtypes = [ctypes.c_void_p, c_double_p, c_int_p, ctypes.c_int, ctypes.c_int]
    L.mcp_ba_add_meas.argtypes = [ctypes.c_void_p, c_int_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_double, ctypes.c_int]
    L.mcp_ba_add_points.argtypes = [ctypes.c_void_p, ctypes.c_int, c_double_p, c_int_p, ctypes.c_int, c_int_p, c_ubyte_p, c_int_p]
    L.mcp_ba_add_measurements.argtypes = [ctypes.c_void_p, ctypes.c_int, c_int_p, ctypes.c_int, c_int_p, c_int_p, c_double_p, c_double_p, c_int_p]
    L.mcp_ba_compute.argtypes = [ctypes.c_void_p, c_ubyte_p, ctypes.c_int, ctypes.c_double]
    for f in ("mcp_ba_converged", "mcp_ba_total_iterations", "mcp_ba_num_outliers", "mcp_ba_num_iter_logs", "mcp_ba_prepare"):
        getattr(L, f).argtypes = [ctypes.c_void_p]
    for f in ("mcp_ba_sigma_squared", "mcp_ba_mean_chi_squared", "mcp_ba_max_cov", "mcp_ba_lambda"):
        getattr(L, f).argtypes = [ctypes.c_void_p]
        getattr(L, f).restype = ctypes.c_double
    L.mcp_ba_get_point.argtypes = [ctypes.c_void_p, ctypes.c_int, c_double_p]
    L.mcp_ba_get_pose.argtypes = [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p]
    L.mcp_ba_get_points.argtypes = [ctypes.c_void_p, ctypes.c_int, c_int_p, c_double_p]
    L.mcp_ba_get_poses.argtypes = [ctypes.c_void_p, ctypes.c_int, c_int_p, c_double_p, c_double_p]
    L.mcp_ba_get_outliers.argtypes = [ctypes.c_void_p, c_int_p, ctypes.c_int]
    L.mcp_ba_get_iter_logs.argtypes = [ctypes.c_void_p, ctypes.POINTER(McpBaIterLog), ctypes.c_int]
    L.mcp_ba_get_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(McpBaTiming)]
    L.mcp_ba_set_allreduce.argtypes = [ctypes.c_void_p, ALLREDUCE_FN, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.mcp_ba_eval.argtypes = [ctypes.c_void_p, c_double_p, c_double_p]
    L.mcp_ba_robust_chi2.argtypes = [ctypes.c_void_p, c_double_p, c_double_p]
    L.mcp_ba_debug_solve.argtypes = [ctypes.c_void_p, ctypes.c_double, c_double_p]
    L.mcp_dense_spd_solve.argtypes = [c_double_p, ctypes.c_int, c_double_p, c_double_p]
    L.mcp_chol_debug_factor.argtypes = [c_double_p, ctypes.c_int, c_double_p, c_double_p, c_double_p, ctypes.POINTER(ctypes.c_int)]
    L.mcp_debug_pose_cut.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.mcp_chol_time.argtypes = [c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p]
    L.mcp_dense_spd_stress.argtypes = [c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.POINTER(ctypes.c_int)]
    L.mcp_ba_debug_system.argtypes = [ctypes.c_void_p, ctypes.c_double, c_double_p]
    L.mcp_comm_unique_id.argtypes = [ctypes.c_void_p]
    L.mcp_comm_init.restype = ctypes.c_void_p
    L.mcp_comm_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.mcp_comm_destroy.argtypes = [ctypes.c_void_p]
    L.mcp_ba_set_comm.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.mcp_comm_allreduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.mcp_comm_allreduce_lane.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    _LIB = L
    return L


def last_error():
    return lib().mcp_last_error().decode()


def device_count():
    return lib().mcp_device_count()


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


COMM_ID_BYTES = 128


def comm_unique_id():
    """RCCL unique id (rank 0 creates it and ships the bytes to the other ranks)."""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    if lib().mcp_comm_unique_id(buf) != 0:
        raise RuntimeError("mcp_comm_unique_id failed: " + last_error())
    return bytes(buf.raw)


class Comm:
    """RCCL communicator over the GPUs of one node (include/mcp_ba.h mcp_comm_*)."""

    def __init__(self, uid, rank, world_size, device=-1):
        self._L = lib()
        self._h = self._L.mcp_comm_init(ctypes.c_char_p(uid), int(rank), int(world_size), int(device))
        if not self._h:
            raise RuntimeError("mcp_comm_init failed: " + last_error())
        self.rank, self.world_size = rank, world_size

    def allreduce(self, device_ptr, count, lane=0):
        """lane 0: the RCCL communicator of the solver's main stream, 1: the speculative stream's (ncclCommSplit of lane 0)"""
        if self._L.mcp_comm_allreduce_lane(self._h, ctypes.c_void_p(device_ptr), int(count), int(lane)) != 0:
            raise RuntimeError("mcp_comm_allreduce failed: " + last_error())

    def close(self):
        if getattr(self, "_h", None):
            self._L.mcp_comm_destroy(self._h)
            self._h = None


def dense_spd_solve(A, b):
    """Solve A x = b on the GPU with the reduced-system Cholesky kernels (test hook)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b)
    if lib().mcp_dense_spd_solve(_dp(A), A.shape[0], _dp(b), _dp(x)) != 0:
        raise RuntimeError("mcp_dense_spd_solve failed: " + last_error())
    return x


def struct_cache_stats():
    """(hits, misses) of the structure cache so far in this process (include/mcp_ba.h)."""
    h, m = ctypes.c_longlong(0), ctypes.c_longlong(0)
    L = lib()
    L.mcp_ba_struct_cache_stats.restype = None
    L.mcp_ba_struct_cache_stats(ctypes.byref(h), ctypes.byref(m))
    return int(h.value), int(m.value)


def struct_cache_near_hits():
    """Prepare() calls so far that adopted the cached structure of a SUPERSET of their measurements (include/mcp_ba.h, near miss)."""
    L = lib()
    L.mcp_ba_struct_cache_near_hits.restype = ctypes.c_longlong
    return int(L.mcp_ba_struct_cache_near_hits())


def struct_cache_clear():
    L = lib()
    L.mcp_ba_struct_cache_clear.restype = None
    L.mcp_ba_struct_cache_clear()


def chol_debug_factor(A, b):
    """The one-launch factorisation seen from outside (test hook): (L with L_kk^-1 in its diagonal 32x32 blocks, y = L^-1 b,
    hand-off error word, failure flag)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = A.shape[0]
    L = np.zeros((n, n)); y = np.zeros(n)
    info = (ctypes.c_int * 2)(0, 0)
    if lib().mcp_chol_debug_factor(_dp(A), n, _dp(b), _dp(L), _dp(y), info) != 0:
        raise RuntimeError("mcp_chol_debug_factor failed: " + last_error())
    return L, y, info[0], info[1]


def pose_cut(adjacency, max_arcs=6, threads=1):
    """The cut of a pose coupling graph (nf x nf, nonzero = coupled; free poses in add order) that Prepare() gives the factorisation its
    chains with (csrc/ba_cut.h; host code, no GPU).  Returns dict(order, segs, found, taken, relabelled, arcs, opened_at, steps,
    steps_one_chain, separator, arc_len)."""
    A = np.ascontiguousarray(np.asarray(adjacency) != 0, dtype=np.uint8)
    nf = A.shape[0]
    order = (ctypes.c_int * nf)(); segs = (ctypes.c_int * 8)(); info = (ctypes.c_int * 14)()
    rc = lib().mcp_debug_pose_cut(A.ctypes.data_as(ctypes.c_char_p), nf, int(max_arcs), int(threads), order, segs, info)
    if rc < 0:
        raise RuntimeError("mcp_debug_pose_cut failed: " + last_error())
    return dict(chains=rc, order=np.array(order[:], dtype=np.int64), segs=[v for v in segs[:] if v >= 0], found=bool(info[0]), taken=bool(info[1]),
                relabelled=bool(info[2]), arcs=info[3], opened_at=info[4], steps=info[5], steps_one_chain=info[6], separator=info[7],
                arc_len=[info[8 + i] for i in range(info[3])])


def chol_time(A, b, nsys=1, reps=20, band=0):
    """Device milliseconds per solve of the factorisation and the back-substitution launches (tuning hook) and the solutions."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros((nsys, A.shape[0]))
    tf = ctypes.c_double(0); tb = ctypes.c_double(0)
    if lib().mcp_chol_time(_dp(A), A.shape[0], _dp(b), int(nsys), int(reps), int(band), ctypes.byref(tf), ctypes.byref(tb), _dp(x)) != 0:
        raise RuntimeError("mcp_chol_time failed: " + last_error())
    return tf.value, tb.value, x


def dense_spd_stress(A, b, nsys=1, reps=10):
    """(A + q I) x_q = b for q < nsys, `reps` times from the same device-resident input (test hook).
    Returns (x of the first repetition as (nsys, n), number of repetitions that differed from it in any bit)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros((nsys, A.shape[0]))
    bad = ctypes.c_int(0)
    if lib().mcp_dense_spd_stress(_dp(A), A.shape[0], _dp(b), int(nsys), int(reps), _dp(x), ctypes.byref(bad)) != 0:
        raise RuntimeError("mcp_dense_spd_stress failed: " + last_error())
    return x, bad.value


class ChainBundle:
    """ChainBundle(cameraModels, bUseRobust, bUseTukey, bVerbose) -- ChainBundle.h:106."""

    snMaxIterations = 100
    snMaxTrialsAfterFailure = 100
    sdUpdatePercentConvergenceLimit = 1e-10
    sdUpdateRMSConvergenceLimit = 1e-10
    sdMinMEstimatorSigma = 0.5

    def __init__(self, cams, use_robust=True, use_tukey=True, verbose=False, disable_convergence=False,
                 device=-1, profile=False):
        self._L = lib()
        self._cams = camera_array(cams)
        prm = McpBaParams(self.snMaxIterations, self.snMaxTrialsAfterFailure, self.sdUpdatePercentConvergenceLimit,
                          self.sdUpdateRMSConvergenceLimit, self.sdMinMEstimatorSigma, int(disable_convergence),
                          int(device), int(profile))
        t0 = time.perf_counter()
        self._h = self._L.mcp_ba_create(ctypes.cast(self._cams, ctypes.c_void_p), len(cams), int(use_robust),
                                        int(use_tukey), int(verbose), ctypes.byref(prm))
        self.abi_create_seconds = time.perf_counter() - t0
        if not self._h:
            raise RuntimeError("mcp_ba_create failed: " + last_error())
        self.abort = ctypes.c_ubyte(0)      # the caller's mbBundleAbortRequested
        self._hook = None
        self.abi_seconds = 0.0               # time spent inside the library's Add* entries (what a native caller pays; the rest is Python)
        self.abi_read_seconds = 0.0          # ... and inside its Get* entries

    def close(self):
        if getattr(self, "_h", None):
            self._L.mcp_ba_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc < 0:
            raise RuntimeError("%s failed: %s" % (what, last_error()))
        return rc

    def AddPose(self, R, t, fixed):
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
        t = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
        return self._L.mcp_ba_add_pose(self._h, _dp(R), _dp(t), int(fixed))

    def AddPoint(self, x, chain, fixed):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(3)
        c = np.ascontiguousarray(chain, dtype=np.int32)
        return self._check(self._L.mcp_ba_add_point(self._h, _dp(x), _ip(c), len(c), int(fixed)), "AddPoint")

    def AddMeas(self, chain, point_id, uv, sigma_sq, cam_index):
        c = np.ascontiguousarray(chain, dtype=np.int32)
        uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(2)
        self._check(self._L.mcp_ba_add_meas(self._h, _ip(c), len(c), int(point_id), _dp(uv), float(sigma_sq), int(cam_index)), "AddMeas")

    def AddPointBatch(self, x, chains, chain_len, fixed):
        x = np.ascontiguousarray(x, dtype=np.float64)
        chains = np.ascontiguousarray(chains, dtype=np.int32)
        chain_len = np.ascontiguousarray(chain_len, dtype=np.int32)
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        ids = np.zeros(x.shape[0], dtype=np.int32)
        t0 = time.perf_counter()
        rc = self._L.mcp_ba_add_points(self._h, x.shape[0], _dp(x), _ip(chains), chains.shape[1], _ip(chain_len),
                                       fixed.ctypes.data_as(c_ubyte_p), _ip(ids))
        self.abi_seconds += time.perf_counter() - t0
        self._check(rc, "AddPointBatch")
        return ids

    def AddMeasBatch(self, chains, chain_len, point_ids, uv, sigma_sq, cam_index):
        chains = np.ascontiguousarray(chains, dtype=np.int32)
        chain_len = np.ascontiguousarray(chain_len, dtype=np.int32)
        point_ids = np.ascontiguousarray(point_ids, dtype=np.int32)
        uv = np.ascontiguousarray(uv, dtype=np.float64)
        sigma_sq = np.ascontiguousarray(sigma_sq, dtype=np.float64)
        cam_index = np.ascontiguousarray(cam_index, dtype=np.int32)
        t0 = time.perf_counter()
        rc = self._L.mcp_ba_add_measurements(self._h, uv.shape[0], _ip(chains), chains.shape[1], _ip(chain_len),
                                             _ip(point_ids), _dp(uv), _dp(sigma_sq), _ip(cam_index))
        self.abi_seconds += time.perf_counter() - t0
        self._check(rc, "AddMeasBatch")

    def DebugSystem(self, lam):
        """(S, rhs, J^T r) of the reduced pose system at the current state (test hook, mcp_ba_debug_system)."""
        n = self._L.mcp_ba_debug_system(self._h, float(lam), None)
        if n < 0:
            raise RuntimeError("mcp_ba_debug_system: " + last_error())
        out = np.zeros(n * n + 2 * n)
        if self._L.mcp_ba_debug_system(self._h, float(lam), _dp(out)) < 0:
            raise RuntimeError("mcp_ba_debug_system: " + last_error())
        return out[:n * n].reshape(n, n), out[n * n:n * n + n], out[n * n + n:]

    def Compute(self, n_iter=None, user_lambda=-1.0):
        """int Compute(bool* pAbortSignal, int nNumIter, double dUserLambda); the abort flag is self.abort."""
        n = self.snMaxIterations if n_iter is None else int(n_iter)
        rc = self._L.mcp_ba_compute(self._h, ctypes.byref(self.abort), n, float(user_lambda))
        if rc == -2:        # MCP_BA_ERR_RUNTIME: a HIP / RCCL call failed -- not one of the reference's outcomes
            raise RuntimeError("mcp_ba_compute: " + last_error())
        return rc

    def Converged(self):
        return bool(self._L.mcp_ba_converged(self._h))

    def TotalIterations(self):
        return self._L.mcp_ba_total_iterations(self._h)

    def GetPoint(self, pid):
        x = np.zeros(3)
        self._check(self._L.mcp_ba_get_point(self._h, int(pid), _dp(x)), "GetPoint")
        return x

    def GetPose(self, pid):
        R = np.zeros(9)
        t = np.zeros(3)
        self._check(self._L.mcp_ba_get_pose(self._h, int(pid), _dp(R), _dp(t)), "GetPose")
        return R.reshape(3, 3), t

    def GetPoints(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        x = np.zeros((len(ids), 3))
        t0 = time.perf_counter()
        rc = self._L.mcp_ba_get_points(self._h, len(ids), _ip(ids), _dp(x))
        self.abi_read_seconds += time.perf_counter() - t0
        self._check(rc, "GetPoints")
        return x

    def GetPoses(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        R = np.zeros((len(ids), 3, 3))
        t = np.zeros((len(ids), 3))
        t0 = time.perf_counter()
        rc = self._L.mcp_ba_get_poses(self._h, len(ids), _ip(ids), _dp(R), _dp(t))
        self.abi_read_seconds += time.perf_counter() - t0
        self._check(rc, "GetPoses")
        return R, t

    def GetOutlierMeasurements(self):
        t0 = time.perf_counter()
        n = self._L.mcp_ba_num_outliers(self._h)
        out = np.empty((max(n, 1), 3), dtype=np.int32)
        n = self._L.mcp_ba_get_outliers(self._h, _ip(out), n)
        self.abi_read_seconds += time.perf_counter() - t0
        return [tuple(int(v) for v in out[i]) for i in range(n)]

    def GetSigmaSquared(self):
        return self._L.mcp_ba_sigma_squared(self._h)

    def GetMeanChiSquared(self):
        return self._L.mcp_ba_mean_chi_squared(self._h)

    def GetMaxCov(self):
        return self._L.mcp_ba_max_cov(self._h)

    def GetLambda(self):
        return self._L.mcp_ba_lambda(self._h)

    def IterLogs(self):
        n = self._L.mcp_ba_num_iter_logs(self._h)
        arr = (McpBaIterLog * max(n, 1))()
        n = self._L.mcp_ba_get_iter_logs(self._h, arr, n)
        return [dict(chi2_start=a.chi2_start, chi2_end=a.chi2_end, lambda_end=a.lambda_end, sigma_sq=a.sigma_sq,
                     rms_update=a.rms_update, trials=a.trials, accepted=a.accepted) for a in arr[:n]]

    def Timing(self):
        t = McpBaTiming()
        self._L.mcp_ba_get_timing(self._h, ctypes.byref(t))
        return {f: getattr(t, f) for f, _ in McpBaTiming._fields_}

    def SetAllReduce(self, fn, rank, world_size):
        """fn(device_ptr:int, count:int, stream:int) -> None performs an in-place SUM all-reduce of doubles."""
        def tramp(_user, buf, count, stream):
            try:
                fn(int(buf), int(count), int(stream or 0))
                return 0
            except Exception as exc:      # never let an exception cross the C ABI
                print("all-reduce hook raised:", repr(exc), flush=True)
                return 1
        self._hook = ALLREDUCE_FN(tramp) if fn is not None else ctypes.cast(None, ALLREDUCE_FN)
        self._check(self._L.mcp_ba_set_allreduce(self._h, self._hook, None, int(rank), int(world_size)), "SetAllReduce")

    def SetComm(self, comm):
        """Use a native RCCL communicator (stream-ordered all-reduces) instead of a host hook."""
        self._comm = comm
        self._check(self._L.mcp_ba_set_comm(self._h, comm._h if comm is not None else None), "SetComm")

    # ---- introspection for parity tests ----
    def Prepare(self):
        return self._check(self._L.mcp_ba_prepare(self._h), "Prepare")

    def Eval(self, n_meas):
        chi2 = np.zeros(n_meas)
        err = np.zeros((n_meas, 2))
        self._check(self._L.mcp_ba_eval(self._h, _dp(chi2), _dp(err)), "Eval")
        return chi2, err

    def DebugRobustChi2(self):
        s = ctypes.c_double(0)
        c = ctypes.c_double(0)
        self._check(self._L.mcp_ba_robust_chi2(self._h, ctypes.byref(s), ctypes.byref(c)), "RobustChi2")
        return c.value, s.value

    def DebugSolve(self, lam):
        n = self.Prepare()
        x = np.zeros(n)
        self._check(self._L.mcp_ba_debug_solve(self._h, float(lam), _dp(x)), "DebugSolve")
        return x
