"""Bring-up of the one-launch factorisation (ba_chol2.h) on the GPU: L / L_kk^-1 / y against numpy per size, the solve, timing.
Usage: python scripts/gpu_chol2.py [sizes...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcptam_amd import chain_bundle as cb


def spd(n, seed=0, band=0):
    rng = np.random.default_rng(seed + n)
    B = rng.normal(size=(n, n))
    A = B @ B.T + n * np.eye(n)
    if band:
        ntc = (n + 31) // 32
        for i in range(ntc):
            for j in range(ntc):
                lo, hi = max(i, j), min(i, j)
                if not (lo - hi <= band or lo >= ntc - band):
                    A[32 * i:32 * i + 32, 32 * j:32 * j + 32] = 0.0
        A += 4 * n * np.eye(n)
    return A, rng.normal(size=n)


def dissect(n, b=6, seed=0):
    """A pure band of b tiles (n a multiple of 32), block rows / columns re-ordered [left half ascending | right half DEscending |
    the b blocks between them]: two independent chains and a border (what mcp_chol_time's band = -b plan expects)."""
    assert n % 32 == 0
    ntc = n // 32
    rng = np.random.default_rng(seed + n)
    B = rng.normal(size=(n, n))
    A = B @ B.T
    for i in range(ntc):
        for j in range(ntc):
            if abs(i - j) > b:
                A[32 * i:32 * i + 32, 32 * j:32 * j + 32] = 0.0
    A += 5 * n * np.eye(n)
    h = (ntc - b) // 2
    # original order: [left 0..h-1][separator h..h+b-1][right h+b..ntc-1]
    order = list(range(h)) + list(range(ntc - 1, h + b - 1, -1)) + list(range(h, h + b))
    idx = np.concatenate([np.arange(32 * o, 32 * o + 32) for o in order])
    return A[np.ix_(idx, idx)], rng.normal(size=n)


def check_factor(n):
    A, b = spd(n)
    L, y, err, fail = cb.chol_debug_factor(np.tril(A), b)
    Lr = np.linalg.cholesky(A)
    yr = np.linalg.solve(Lr, b)
    ntc = (n + 31) // 32
    eo, ed = 0.0, 0.0
    for i in range(ntc):
        for j in range(i + 1):
            g = L[32 * i:32 * i + 32, 32 * j:32 * j + 32]
            r = Lr[32 * i:32 * i + 32, 32 * j:32 * j + 32]
            if i == j:
                ed = max(ed, np.abs(g - np.linalg.inv(r)).max() / np.abs(np.linalg.inv(r)).max())
            else:
                e = np.abs(g - r).max() / max(np.abs(r).max(), 1e-300)
                if e > 1e-9 and eo <= 1e-9:
                    print("    first bad off-diagonal tile (%d,%d): %.3e" % (i, j, e))
                eo = max(eo, e)
    ey = np.abs(y - yr).max() / np.abs(yr).max()
    print("factor n=%5d  err=0x%x fail=%d  offdiag %.2e  diag-inverse %.2e  y %.2e" % (n, err, fail, eo, ed, ey), flush=True)
    return err == 0 and fail == 0 and max(eo, ed, ey) < 1e-10


def check_solve(n):
    A, b = spd(n)
    x = cb.dense_spd_solve(np.tril(A), b)
    r = np.linalg.solve(A, b)
    e = np.abs(x - r).max() / np.abs(r).max()
    print("solve  n=%5d  %.2e" % (n, e), flush=True)
    return e < 1e-11


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [1, 5, 31, 32, 33, 64, 65, 96, 97, 100, 130, 200, 500, 1194]
    ok = True
    for n in sizes:
        t0 = time.time()
        try:
            ok &= check_factor(n)
            ok &= check_solve(n)
        except Exception as exc:
            print("n=%d: %r" % (n, exc), flush=True)
            ok = False
        if time.time() - t0 > 30:
            print("slow: giving up"); break
    print("ALL OK" if ok else "FAILURES", flush=True)
    for n, nsys, band in ((1194, 1, 0), (1194, 1, 6), (1194, 3, 6), (2994, 1, 0)):
        A, b = spd(n, band=band)
        try:
            tf, tb, x = cb.chol_time(np.tril(A), b, nsys=nsys, reps=30, band=band)
            e = max(np.abs(x[q] - np.linalg.solve(A + q * np.eye(n), b)).max() / np.abs(x[q]).max() for q in range(nsys))
            print("time n=%d nsys=%d band=%d: factor %.1f us  back %.1f us  (rel err %.1e)  persist=%s" % (n, nsys, band, tf * 1e3, tb * 1e3, e, os.environ.get("MCP_BA_CHOL_PERSIST", "1")), flush=True)
        except Exception as exc:
            print("time n=%d nsys=%d band=%d: %r" % (n, nsys, band, exc), flush=True)
    for n, nsys, b in ((1216, 1, 6), (1216, 4, 6), (3008, 1, 6)):
        A, rhs = dissect(n, b)
        A0, rhs0 = spd(n, band=b)
        try:
            tf, tb, x = cb.chol_time(np.tril(A), rhs, nsys=nsys, reps=30, band=-b)
            e = max(np.abs(x[q] - np.linalg.solve(A + q * np.eye(n), rhs)).max() / np.abs(x[q]).max() for q in range(nsys))
            tf0, tb0, _ = cb.chol_time(np.tril(A0), rhs0, nsys=nsys, reps=30, band=b)
            print("dissected n=%d nsys=%d band=%d: factor %.1f us  back %.1f us  (rel err %.1e)   | one chain (banded + bordered): factor %.1f us  back %.1f us" % (n, nsys, b, tf * 1e3, tb * 1e3, e, tf0 * 1e3, tb0 * 1e3), flush=True)
        except Exception as exc:
            print("dissected n=%d nsys=%d band=%d: %r" % (n, nsys, b, exc), flush=True)
