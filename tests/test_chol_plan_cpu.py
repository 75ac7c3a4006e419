"""Host logic of the block-sparse Cholesky plan (mcptam_amd/csrc/ba_chol.h, CholPlan::build): the symbolic fill and the
per-step tile schedule, replayed in numpy with the launch semantics of k_chol_step (every tile of a step reads the state the
previous step left) and compared with a dense factorisation.  Runs without a GPU: only the host-side vectors are used."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB = 32


@pytest.fixture(scope="module")
def dump_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("plan") / "chol_plan_dump")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950",
                           os.path.join(ROOT, "tests", "cpp", "chol_plan_dump.hip"), "-o", exe])
    return exe


def _plan(exe, n, pattern):
    ntc = pattern.shape[0]
    text = "%d %d\n" % (n, ntc) + "\n".join(" ".join(str(int(v)) for v in row) for row in pattern) + "\n"
    out = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    n_, ntc_, ntr = (int(v) for v in lines[0].split())
    assert (n_, ntc_) == (n, ntc)
    steps, rows = [], []
    for ln in lines[1:]:
        head, _, rest = ln.partition(":")
        items = rest.split()
        if head.startswith("step"):
            steps.append([tuple(int(v) for v in it.split(",")) for it in items])
        else:
            rows.append([int(v) for v in items])
    return ntr, steps, rows


def _symbolic_fill(ntc, ntr, pattern, rhs_tile):
    """Textbook symbolic Cholesky on the tile graph (rows ntc..ntr-1 only carry the right-hand side)."""
    P = np.zeros((ntr, ntc), dtype=bool)
    P[:ntc, :ntc] = np.tril(pattern.astype(bool))
    P[np.arange(ntc), np.arange(ntc)] = True
    P[rhs_tile, :] = True
    for k in range(ntc):
        below = [i for i in range(k + 1, ntr) if P[i, k]]
        for a in below:
            for b in below:
                if b <= a and b < ntc:
                    P[a, b] = True
    return P


def _replay(n, ntr, steps, S, rhs):
    """The arithmetic of k_chol_step at tile granularity; A is (n+1) x n with the right-hand side as row n."""
    A = np.zeros((ntr * NB, ntr * NB))
    A[:n, :n] = np.tril(S)
    A[n, :n] = rhs
    for k, tiles in enumerate(steps):
        before = A.copy()           # one launch: every workgroup reads what the previous launch left
        k0, p0 = k * NB, (k - 1) * NB
        nbe = min(NB, n - k0)
        for (ti, tj) in tiles:
            r0, c0 = ti * NB, tj * NB
            C = before[r0:r0 + NB, c0:c0 + NB].copy()
            if k > 0:
                C -= before[r0:r0 + NB, p0:p0 + NB] @ before[c0:c0 + NB, p0:p0 + NB].T
            if tj != k:
                A[r0:r0 + NB, c0:c0 + NB] = np.tril(C) if ti == tj else C
                continue
            D = before[k0:k0 + NB, k0:k0 + NB].copy()
            if k > 0:
                D -= before[k0:k0 + NB, p0:p0 + NB] @ before[k0:k0 + NB, p0:p0 + NB].T
            Lkk = np.linalg.cholesky(D[:nbe, :nbe])          # reads the lower triangle only, like the panel kernel
            if ti == k:
                A[k0:k0 + nbe, k0:k0 + nbe] = Lkk
                lo = C[nbe:, :nbe]                     # rows beyond the matrix in the diagonal tile: the right-hand side
                if lo.size:
                    A[k0 + nbe:k0 + NB, k0:k0 + nbe] = np.linalg.solve(Lkk, lo.T).T
            else:
                A[r0:r0 + NB, k0:k0 + nbe] = np.linalg.solve(Lkk, C[:, :nbe].T).T
    return A


def _random_pattern(rng, ntc, kind):
    P = np.eye(ntc, dtype=int)
    if kind == "dense":
        P[:] = 1
    elif kind == "chain":          # a trajectory: neighbours see the same points
        for i in range(ntc):
            for j in range(max(0, i - 2), i):
                P[i, j] = 1
    else:                          # chain + loop closures
        for i in range(1, ntc):
            P[i, i - 1] = 1
        for _ in range(ntc):
            i, j = sorted(rng.integers(0, ntc, 2))
            P[j, i] = 1
    return np.tril(P)


@pytest.mark.parametrize("n,kind", [(70, "dense"), (96, "chain"), (200, "loops"), (333, "chain"), (417, "loops"), (640, "dense")])
def test_plan_schedule_reproduces_dense_cholesky(dump_exe, n, kind):
    rng = np.random.default_rng(n)
    ntc = (n + NB - 1) // NB
    pattern = _random_pattern(rng, ntc, kind)
    ntr, steps, rows = _plan(dump_exe, n, pattern)
    assert ntr == (n + 1 + NB - 1) // NB
    rhs_tile = n // NB
    fill = _symbolic_fill(ntc, ntr, pattern, rhs_tile)
    # the first entries of a step are its block column (the critical path), each structurally non-zero tile exactly once
    for k, tiles in enumerate(steps):
        col = [t for t in tiles if t[1] == k]
        assert tiles[:len(col)] == col and col[0] == (k, k)
        assert sorted(t[0] for t in col) == [i for i in range(k, ntr) if fill[i, k]]
        assert len(set(tiles)) == len(tiles)
        for (ti, tj) in tiles:
            assert fill[ti, tj] and ti >= tj >= k
    for r in range(ntc):
        assert rows[r] == [j for j in range(r) if fill[r, j]]
    # a matrix with exactly that tile sparsity
    M = np.zeros((n, n))
    for i in range(ntc):
        for j in range(i + 1):
            if pattern[i, j]:
                blk = rng.normal(size=(NB, NB))
                r1, c1 = min(n, (i + 1) * NB), min(n, (j + 1) * NB)
                M[i * NB:r1, j * NB:c1] = blk[:r1 - i * NB, :c1 - j * NB]
    S = np.tril(M) + np.tril(M, -1).T
    S += np.eye(n) * (np.abs(S).sum(axis=1).max() + 1.0)
    rhs = rng.normal(size=n)
    A = _replay(n, ntr, steps, S, rhs)
    L = np.linalg.cholesky(S)
    assert np.abs(A[:n, :n] - L).max() < 1e-9 * np.abs(L).max()
    assert np.abs(A[n, :n] - np.linalg.solve(L, rhs)).max() < 1e-9 * np.abs(rhs).max()
    # nothing outside the symbolic fill is touched
    Lt = np.abs(L).reshape(-1)
    for i in range(ntc):
        for j in range(i + 1):
            if not fill[i, j]:
                assert np.abs(L[i * NB:(i + 1) * NB, j * NB:(j + 1) * NB]).max() < 1e-12 * Lt.max()


def test_replay_notices_a_missing_tile(dump_exe):
    """The replay above is a real check: dropping one trailing tile from the schedule breaks the factorisation."""
    n = 200
    rng = np.random.default_rng(7)
    ntc = (n + NB - 1) // NB
    pattern = _random_pattern(rng, ntc, "dense")
    ntr, steps, _ = _plan(dump_exe, n, pattern)
    M = rng.normal(size=(n, n))
    S = M @ M.T + n * np.eye(n)
    rhs = rng.normal(size=n)
    L = np.linalg.cholesky(S)
    assert np.abs(_replay(n, ntr, steps, S, rhs)[:n, :n] - L).max() < 1e-9 * np.abs(L).max()
    broken = [list(t) for t in steps]
    victim = [t for t in broken[2] if t[1] != 2][-1]
    broken[2].remove(victim)
    assert np.abs(_replay(n, ntr, broken, S, rhs)[:n, :n] - L).max() > 1e-3 * np.abs(L).max()
