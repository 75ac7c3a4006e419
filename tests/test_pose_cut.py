"""The cut of the pose coupling graph that gives the one-launch factorisation its chains (mcptam_amd/csrc/ba_cut.h) -- host code,
reached through the debug hook mcp_debug_pose_cut of the C ABI: no GPU needed.  What Prepare() relies on: the order is a permutation, poses of
different arcs do not couple (the plan would refuse such chains), every arc but the last ends on a tile boundary, the result does not depend on
how the search is split over threads, a graph without a small separator is left alone, an add order that is not the trajectory's is relabelled."""
import numpy as np
import pytest


def band_graph(nf, reach, ring):
    i = np.arange(nf)
    d = np.abs(i[:, None] - i[None, :])
    if ring:
        d = np.minimum(d, nf - d)
    return (d <= reach) & (d > 0)


def with_chords(A, rng, clusters=10, width=6):
    """groups of `width` poses that see the same far points from across the loop (the BASELINE generator's small loop does that)"""
    A = A.copy()
    nf = A.shape[0]
    for _ in range(clusters):
        a = int(rng.integers(0, nf)); b = (a + int(rng.integers(nf // 4, nf // 2))) % nf
        ia = (a + np.arange(width)) % nf; ib = (b + np.arange(width)) % nf
        A[np.ix_(ia, ib)] = True; A[np.ix_(ib, ia)] = True
    np.fill_diagonal(A, False)
    return A


def arcs_of(res, nf):
    """poses of every arc and of the separator, from the order and the arc lengths"""
    out, p0 = [], 0
    for n in res["arc_len"]:
        out.append(res["order"][p0:p0 + n]); p0 += n
    return out, res["order"][p0:]


def check_cut(A, res):
    nf = A.shape[0]
    assert sorted(res["order"].tolist()) == list(range(nf))
    if not res["taken"]:
        assert res["chains"] == 1 and res["segs"] == []
        return
    arcs, sep = arcs_of(res, nf)
    assert res["chains"] == len(arcs) + 1 == len(res["segs"])
    assert sum(len(a) for a in arcs) + len(sep) == nf and len(sep) == res["separator"]
    for i, a in enumerate(arcs):
        for j, b in enumerate(arcs):
            if i < j:
                assert not A[np.ix_(a, b)].any(), "arcs %d and %d couple" % (i, j)
    cum = 0
    for i, a in enumerate(arcs):
        if i + 1 < len(arcs):
            assert len(a) % 16 == 0, "an arc before the last must end on a tile boundary (16 poses = 3 tiles)"
        assert res["segs"][i] == 6 * cum // 32
        cum += len(a)
        assert 6 * cum // 32 - res["segs"][i] >= 3
    t_all = (6 * nf + 31) // 32
    assert res["steps_one_chain"] == t_all
    assert 10 * res["steps"] <= 8 * t_all


@pytest.mark.parametrize("nf,reach,ring", [(199, 5, True), (199, 5, False), (130, 4, True), (499, 6, True), (96, 3, False), (1024, 5, True)])
def test_bands_and_rings_are_cut_into_arcs_that_do_not_couple(nf, reach, ring):
    from mcptam_amd import chain_bundle
    A = band_graph(nf, reach, ring)
    res = chain_bundle.pose_cut(A)
    assert res["found"] and res["taken"] and not res["relabelled"]
    check_cut(A, res)
    if not ring:
        assert res["opened_at"] == 0


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_far_couplings_go_to_the_separator(seed):
    from mcptam_amd import chain_bundle
    rng = np.random.default_rng(seed)
    A = with_chords(band_graph(199, 12, True), rng)
    res = chain_bundle.pose_cut(A)
    check_cut(A, res)
    assert res["taken"]
    # (two arcs at most: same promise)
    res2 = chain_bundle.pose_cut(A, max_arcs=2)
    check_cut(A, res2)
    assert res2["arcs"] == 2 and res2["steps"] >= res["steps"]


def test_the_cut_does_not_depend_on_the_number_of_threads():
    from mcptam_amd import chain_bundle
    rng = np.random.default_rng(7)
    for A in (with_chords(band_graph(199, 12, True), rng), band_graph(499, 6, True), band_graph(150, 5, False)):
        ref = chain_bundle.pose_cut(A, threads=1)
        for t in (2, 3, 8, 16, 64):
            r = chain_bundle.pose_cut(A, threads=t)
            assert np.array_equal(r["order"], ref["order"]) and r["segs"] == ref["segs"] and r["steps"] == ref["steps"]


def test_a_graph_without_a_small_separator_is_left_alone():
    from mcptam_amd import chain_bundle
    rng = np.random.default_rng(11)
    nf = 160
    A = rng.random((nf, nf)) < 0.3
    A = A | A.T
    np.fill_diagonal(A, False)
    res = chain_bundle.pose_cut(A)
    assert not res["taken"] and res["chains"] == 1
    assert np.array_equal(res["order"], np.arange(nf)) or res["relabelled"]
    check_cut(A, res)
    full = np.ones((120, 120), bool); np.fill_diagonal(full, False)
    res = chain_bundle.pose_cut(full)
    assert not res["taken"] and np.array_equal(res["order"], np.arange(120))


@pytest.mark.parametrize("ring", [True, False])
def test_an_add_order_that_is_not_the_trajectory_is_relabelled_first(ring):
    """MCPTAM hands its key frames over in the order of a std::set of pointers: the same graph with its poses numbered at random."""
    from mcptam_amd import chain_bundle
    rng = np.random.default_rng(5)
    nf = 199
    A = band_graph(nf, 5, ring)
    perm = rng.permutation(nf)
    B = A[np.ix_(perm, perm)]
    res = chain_bundle.pose_cut(B)
    assert res["relabelled"] and res["taken"]
    check_cut(B, res)
    # the trajectory's own order is left as it is
    assert not chain_bundle.pose_cut(A)["relabelled"]


def test_the_synthetic_ring_map_gets_its_chains():
    """The coupling graph of a generated map (the `ring` configuration of the GPU tests): per point the MKFs that see or carry it."""
    from mcptam_amd import chain_bundle, synth
    p = synth.make_config("ring")
    free = np.flatnonzero(~p.base_fixed)
    idx = -np.ones(p.n_mkf, dtype=np.int64); idx[free] = np.arange(len(free))
    nf = len(free)
    M = np.zeros((p.n_points, nf), dtype=bool)
    ok = idx[p.ms_mkf] >= 0
    M[p.ms_pt[ok], idx[p.ms_mkf[ok]]] = True
    src = idx[p.pt_src[:, 0]]
    M[np.arange(p.n_points)[src >= 0], src[src >= 0]] = True
    A = (M.T.astype(np.int32) @ M.astype(np.int32)) > 0
    np.fill_diagonal(A, False)
    res = chain_bundle.pose_cut(A)
    assert res["taken"] and res["chains"] >= 3
    check_cut(A, res)


def test_bad_arguments_are_refused():
    from mcptam_amd import chain_bundle
    with pytest.raises(RuntimeError):
        chain_bundle.pose_cut(np.zeros((1025, 1025), bool))
