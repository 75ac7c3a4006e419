"""Host logic of the one-launch factorisation (mcptam_amd/csrc/ba_chol2.h, CholPersist::build) and the data flow the kernels
run on it, replayed in numpy: the critical workgroup and the helper workgroups are coroutines that may only read a tile (or a
band hand-off) after its owner has published it, scheduled in random order -- so a missing update, a wrong owner or a
circular wait shows up here, without a GPU.  The arithmetic mirrors k_chol_persist / k_chol_back2 step by step."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB = 32


@pytest.fixture(scope="module")
def dump_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("persist") / "chol_persist_dump")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950",
                           os.path.join(ROOT, "tests", "cpp", "chol_persist_dump.hip"), "-o", exe])
    return exe


def _plan(exe, n, pattern, dense=False):
    ntc = pattern.shape[0]
    text = "%d %d %d\n" % (n, ntc, int(dense)) + "\n".join(" ".join(str(int(v)) for v in row) for row in pattern) + "\n"
    out = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    n_, ntc_, nslots, nbslots, nh, W, near = (int(v) for v in lines[0].split())
    assert (n_, ntc_) == (n, ntc)
    slot_of = np.array([[int(v) for v in lines[1 + i].split()] for i in range(ntc + 1)])
    bslot_of = np.array([[int(v) for v in lines[2 + ntc + i].split()] for i in range(ntc + 1)])
    delta_of = [int(v) for v in lines[3 + 2 * ntc].split()]
    helpers = []
    for ln in lines[4 + 2 * ntc:4 + 2 * ntc + nh]:
        head, _, rest = ln.partition(":")
        ti, tj, slot, dslot, kind, in_s, nupd, pre, pre_flag, pre_diag = (int(v) for v in head.split())
        upd = [tuple(int(v) for v in it.split(",")) for it in rest.split()]
        assert len(upd) == nupd
        helpers.append(dict(ti=ti, tj=tj, slot=slot, dslot=dslot, kind=kind, in_s=in_s, upd=upd, pre=pre, pre_flag=pre_flag, pre_diag=pre_diag))
    far = []
    for ln in lines[4 + 2 * ntc + nh:]:
        _, _, rest = ln.partition(":")
        far.append([tuple(int(v) for v in it.split(",")) for it in rest.split()])
    return dict(n=n, ntc=ntc, nslots=nslots, nbslots=nbslots, W=W, near=near, slot_of=slot_of, bslot_of=bslot_of, delta_of=delta_of, helpers=helpers, far=far)


class Wait(Exception):
    pass


def _tile_of(S, rhs, n, ntc, ti, tj):
    """A(ti, tj) as cp_load_A reads it: beyond the matrix a diagonal tile continues as the identity, every other tile as zero; block
    row ntc = the right-hand side in its first row."""
    T = np.zeros((NB, NB))
    c0 = tj * NB
    c1 = min(n, c0 + NB)
    if ti == ntc:
        T[0, :c1 - c0] = rhs[c0:c1]
        return T
    r0 = ti * NB
    r1 = min(n, r0 + NB)
    T[:r1 - r0, :c1 - c0] = S[r0:r1, c0:c1]
    if ti == tj:
        for r in range(r1 - r0, NB):
            T[r, r] = 1.0
    return T


def _potrf(D):
    """cp_potrf: reads the lower triangle of the whole tile (it arrives with the identity beyond the matrix), returns L^-1 (what the
    kernels keep of a diagonal tile)."""
    Dp = np.tril(D) + np.tril(D, -1).T
    return np.linalg.inv(np.linalg.cholesky(Dp))


def _replay_factor(P, S, rhs, rng):
    n, ntc, W = P["n"], P["ntc"], P["W"]
    R = ntc
    slot_of, bslot_of = P["slot_of"], P["bslot_of"]
    Lt, Bt = {}, {}                      # published L tiles / L_kk^-1 by slot; band hand-offs by band slot

    def need(store, key):
        if key not in store:
            raise Wait()
        return store[key]

    def helper(h):
        acc = _tile_of(S, rhs, n, ntc, h["ti"], h["tj"]) if h["in_s"] else np.zeros((NB, NB))
        via_pre = h["kind"] == 1 and h["pre"] >= 0
        for (sa, sb) in (h["upd"][:-1] if via_pre else h["upd"]):
            while True:
                try:
                    A, B = need(Lt, sa), need(Lt, sb)
                    break
                except Wait:
                    yield
            acc = acc - A @ B.T
        if via_pre:
            # last update of a band tile: the far tile of its row is NOT waited for -- its sum before the solve is, and L^-1 of the column
            sa, sb = h["upd"][-1]
            while True:
                try:
                    X = need(Bt, ("pre", h["pre"])) @ need(Lt, h["pre_diag"]).T
                    B = X if sb == sa else need(Lt, sb)
                    break
                except Wait:
                    yield
            acc = acc - X @ B.T
        if h["kind"] == 0 and h["pre"] >= 0:
            Bt[("pre", h["pre"])] = acc.copy()
            yield
        if h["kind"] == 1:
            assert h["dslot"] not in Bt
            Bt[h["dslot"]] = acc
            return
        while h["dslot"] not in Lt:
            yield
        assert h["slot"] not in Lt
        Lt[h["slot"]] = acc @ Lt[h["dslot"]].T

    def critical():
        def band(i, j):
            while bslot_of[i, j] not in Bt:
                yield
            return
        for key in ((0, 0), (1, 0)) + (((1, 1),) if 1 < ntc else ()):
            yield from band(*key)
        Dt = {0: Bt[bslot_of[0, 0]].copy()}
        Tc = Bt[bslot_of[1, 0]].copy()
        if 1 < ntc:
            Dt[1] = Bt[bslot_of[1, 1]].copy()
        Dv = {}
        for s in range(-1, ntc):
            i1, i2 = s + 1, s + 2
            if s >= 0:
                X1 = Tc @ Dv[s].T
                if i1 < ntc:
                    Dt[i1] = Dt[i1] - X1 @ X1.T
            if i1 < ntc:
                Dv[i1] = _potrf(Dt[i1])
            if s < 0:
                continue
            Lt[slot_of[s, s]] = Dv[s]
            Lt[slot_of[i1, s]] = X1
            yield
            if i2 <= R:
                keys = [(i2, s), (i2, i1)] + ([(i2, i2)] if i2 < ntc else [])
                for key in keys:
                    yield from band(*key)
                T2 = Bt[bslot_of[i2, s]]
                Tc = Bt[bslot_of[i2, i1]].copy()
                X2 = T2 @ Dv[s].T
                Tc = Tc - X2 @ X1.T
                if P["delta_of"][i2] >= 0:              # the late product of tile (s+2, s+1), summed with a minus sign by its helper
                    while P["delta_of"][i2] not in Bt:
                        yield
                    Tc = Tc + Bt[P["delta_of"][i2]]
                if i2 < ntc:
                    Dt[i2] = Bt[bslot_of[i2, i2]] - X2 @ X2.T
                Lt[slot_of[i2, s]] = X2
                yield

    agents = [critical()] + [helper(h) for h in P["helpers"]]
    # the launch only promises in-order dispatch: an agent may run once every lower-numbered one has STARTED; give the
    # scheduler a window of resident agents and pick among them at random
    live = list(range(len(agents)))
    window = 24
    idle_rounds = 0
    while live:
        progressed = False
        cand = live[:window]
        rng.shuffle(cand)
        for a in cand:
            before = (len(Lt), len(Bt))
            try:
                next(agents[a])
            except StopIteration:
                live.remove(a)
                progressed = True
                continue
            progressed |= (len(Lt), len(Bt)) != before
        idle_rounds = 0 if progressed else idle_rounds + 1
        assert idle_rounds < 4, "no agent of the resident window can make progress: circular wait (%d left)" % len(live)
    return Lt


def _replay_back(P, Lt):
    n, ntc, near = P["n"], P["ntc"], P["near"]
    slot_of = P["slot_of"]
    x = np.zeros(ntc * NB)
    for k in range(ntc - 1, -1, -1):
        z = Lt[slot_of[ntc, k]][0].copy()
        f = np.zeros(NB)
        for (sl, i) in P["far"][ntc - 1 - k]:
            assert i > k + near and slot_of[i, k] == sl
            f += Lt[sl].T @ x[i * NB:(i + 1) * NB]
        z -= f
        for d in range(near, 0, -1):
            i = k + d
            if i < ntc and slot_of[i, k] >= 0:
                z -= Lt[slot_of[i, k]].T @ x[i * NB:(i + 1) * NB]
        xk = Lt[slot_of[k, k]].T @ z
        xk[min(NB, n - k * NB):] = 0.0
        x[k * NB:(k + 1) * NB] = xk
    # every tile of a column is either near or listed as far
    for k in range(ntc):
        listed = {i for (_, i) in P["far"][ntc - 1 - k]}
        for i in range(k + 1, ntc):
            if slot_of[i, k] >= 0:
                assert (i <= k + near) != (i in listed)
    return x[:n]


def _random_pattern(rng, ntc, kind):
    P = np.eye(ntc, dtype=int)
    if kind == "dense":
        P[:] = 1
    elif kind == "chain":
        for i in range(ntc):
            for j in range(max(0, i - 2), i):
                P[i, j] = 1
    elif kind == "diag":
        pass
    else:
        for i in range(1, ntc):
            P[i, i - 1] = 1
        for _ in range(ntc):
            i, j = sorted(rng.integers(0, ntc, 2))
            P[j, i] = 1
    return np.tril(P)


def _matrix(rng, n, ntc, pattern):
    M = np.zeros((n, n))
    for i in range(ntc):
        for j in range(i + 1):
            if pattern[i, j]:
                blk = rng.normal(size=(NB, NB))
                r1, c1 = min(n, (i + 1) * NB), min(n, (j + 1) * NB)
                M[i * NB:r1, j * NB:c1] = blk[:r1 - i * NB, :c1 - j * NB]
    S = np.tril(M) + np.tril(M, -1).T
    S += np.eye(n) * (np.abs(S).sum(axis=1).max() + 1.0)
    return S


@pytest.mark.parametrize("n,kind", [(1, "dense"), (31, "dense"), (32, "dense"), (33, "dense"), (64, "dense"), (70, "dense"), (96, "chain"), (97, "diag"),
                                    (200, "loops"), (333, "chain"), (417, "loops"), (640, "dense"), (1194, "loops")])
def test_persistent_plan_replays_to_the_dense_solution(dump_exe, n, kind):
    rng = np.random.default_rng(1000 + n)
    ntc = (n + NB - 1) // NB
    pattern = _random_pattern(rng, ntc, kind)
    P = _plan(dump_exe, n, pattern, dense=(kind == "dense"))
    W = P["W"]
    # structure: every tile has exactly one helper, in dependency order; band tiles end their own sums at column i - W
    seen = set()
    key_prev = None
    n_delta = 0
    for h in P["helpers"]:
        ti, tj = h["ti"], h["tj"]
        if ti < 0:                       # the late product of band tile (i, i-1): L(i, i-W) L(i-1, i-W)^T on its own
            i = -1 - ti
            n_delta += 1
            assert tj == i - 1 and h["kind"] == 1 and not h["in_s"] and h["slot"] == P["nslots"] + i and h["dslot"] == P["delta_of"][i]
            assert h["upd"] == [(P["slot_of"][i, i - W], P["slot_of"][i - 1, i - W])]
            key = (i - W, 2, i)
        else:
            assert P["slot_of"][ti, tj] == h["slot"] and (ti, tj) not in seen
            seen.add((ti, tj))
            band = ti - tj < W
            assert h["kind"] == int(band)
            key = (ti - W if band else tj, h["kind"], ti)
            if band and tj == ti - 1 and P["delta_of"][ti] >= 0:
                assert all(sa != P["slot_of"][ti, ti - W] for (sa, _) in h["upd"])      # ... which the tile's own helper leaves out
        assert key_prev is None or key_prev <= key
        key_prev = key
    assert n_delta == sum(1 for d in P["delta_of"] if d >= 0)
    assert len(seen) == P["nslots"] == int((P["slot_of"] >= 0).sum())
    for i in range(ntc):
        for j in range(max(0, i - W + 1), i + 1):
            assert P["slot_of"][i, j] >= 0 and P["bslot_of"][i, j] >= 0       # the band is always there
    assert (P["slot_of"][ntc] >= 0).all()                                       # and so is the right-hand side
    # tiles outside what the assembly writes start from zero, never from S
    S = _matrix(rng, n, ntc, pattern)
    rhs = rng.normal(size=n)
    S_seen = S.copy()
    for h in P["helpers"]:
        if not h["in_s"] and 0 <= h["ti"] < ntc:
            r0, c0 = h["ti"] * NB, h["tj"] * NB
            assert np.abs(S[r0:r0 + NB, c0:c0 + NB]).max() == 0.0
            S_seen[r0:r0 + NB, c0:c0 + NB] = np.nan                              # poison: reading it would show
    Lt = _replay_factor(P, S_seen, rhs, rng)
    L = np.linalg.cholesky(S)
    y = np.linalg.solve(L, rhs)
    scale = np.abs(L).max()
    for i in range(ntc):
        for j in range(i + 1):
            r1, c1 = min(n, (i + 1) * NB), min(n, (j + 1) * NB)
            ref = L[i * NB:r1, j * NB:c1]
            sl = P["slot_of"][i, j]
            if sl < 0:
                assert np.abs(ref).max() < 1e-12 * scale
                continue
            got = Lt[sl][:r1 - i * NB, :c1 - j * NB]
            if i == j:
                assert np.abs(got - np.linalg.inv(ref)).max() < 1e-9 * np.abs(np.linalg.inv(ref)).max()
            else:
                assert np.abs(got - ref).max() < 1e-9 * scale, (i, j)
    for j in range(ntc):
        c1 = min(n, (j + 1) * NB)
        assert np.abs(Lt[P["slot_of"][ntc, j]][0, :c1 - j * NB] - y[j * NB:c1]).max() < 1e-9 * np.abs(y).max()
    x = _replay_back(P, Lt)
    ref = np.linalg.solve(S, rhs)
    assert np.abs(x - ref).max() < 1e-9 * np.abs(ref).max()
