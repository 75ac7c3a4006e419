"""Shared helpers for the parity tests: run one problem through a ChainBundle-shaped object."""
import numpy as np


def run_bundle(bundle, problem, n_iter=None, **kw):
    ids = problem.populate(bundle)
    rc = bundle.Compute(n_iter) if n_iter is not None else bundle.Compute()
    R, t, X = collect(bundle, ids)
    cam = [bundle.GetPose(int(i)) if i > 0 else (None, None) for i in ids["cam"]]
    return dict(cam_R=[a for a, _ in cam], cam_t=[b for _, b in cam], rc=rc, converged=bundle.Converged(), total_iterations=bundle.TotalIterations(),
                sigma_sq=bundle.GetSigmaSquared(), mean_chi2=bundle.GetMeanChiSquared(), max_cov=bundle.GetMaxCov(),
                lam=bundle.GetLambda(), logs=bundle.IterLogs(), outliers=bundle.GetOutlierMeasurements(),
                R=R, t=t, X=X, ids=ids)


def collect(bundle, ids):
    if hasattr(bundle, "GetPoses"):
        R, t = bundle.GetPoses(ids["mkf"])
        X = bundle.GetPoints(ids["point"])
    else:
        Rt = [bundle.GetPose(int(i)) for i in ids["mkf"]]
        R = np.array([a for a, _ in Rt])
        t = np.array([b for _, b in Rt])
        X = np.array([bundle.GetPoint(int(i)) for i in ids["point"]])
    return R, t, X


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def rel_err_elem(a, b, floor_frac=1e-3):
    """Element-wise relative error max_i |a_i - b_i| / max(|b_i|, floor), floor = floor_frac * max|b|: what "within 1e-6 relative"
    (BASELINE.json) means entry by entry; the floor keeps entries that are zero by construction (rotation matrices) from dividing by 0."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    floor = max(floor_frac * float(np.abs(b).max()), 1e-300)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def compare_runs(gpu, ref, tol_state=1e-6, tol_chi=1e-7, allow_iter_slack=0):
    """Compare a HIP run with an oracle run of the same problem.  Returns a report dict; asserts parity."""
    rep = {}
    assert gpu["rc"] == ref["rc"] or abs(gpu["rc"] - ref["rc"]) <= allow_iter_slack, (gpu["rc"], ref["rc"])
    n = min(len(gpu["logs"]), len(ref["logs"]))
    flips = 0
    for i in range(n):
        g, r = gpu["logs"][i], ref["logs"][i]
        if g["trials"] != r["trials"] or g["accepted"] != r["accepted"]:
            flips += 1
            break
        assert abs(g["chi2_start"] - r["chi2_start"]) <= tol_chi * max(abs(r["chi2_start"]), 1e-12) + 1e-18, (i, g, r)
        assert abs(g["lambda_end"] - r["lambda_end"]) <= 1e-6 * abs(r["lambda_end"]), (i, g, r)
    rep["branch_flips"] = flips
    if flips == 0 and gpu["rc"] == ref["rc"]:
        assert gpu["total_iterations"] == ref["total_iterations"]
        assert gpu["converged"] == ref["converged"]
    rep["pose_R"] = rel_err_elem(gpu["R"], ref["R"])
    rep["pose_t"] = rel_err_elem(gpu["t"], ref["t"])
    rep["points"] = rel_err_elem(gpu["X"], ref["X"])
    assert rep["pose_R"] <= tol_state, rep
    assert rep["pose_t"] <= tol_state, rep
    assert rep["points"] <= tol_state, rep
    return rep
