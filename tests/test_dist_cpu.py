"""world_size-2 gloo test of the all-reduce hook used by the sharded solve (runs on CPU)."""
import os
import socket
import tempfile

import numpy as np


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gloo_allreduce_hook_world_size_2():
    import torch.multiprocessing as mp
    import dist_workers
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(dist_workers.hook_on_host_memory, args=(2, _free_port(), d), nprocs=2, join=True)
        r0, r1 = np.load(os.path.join(d, "host_0.npz")), np.load(os.path.join(d, "host_1.npz"))
        expect = np.arange(1000, dtype=np.float64) * 3
        assert np.array_equal(r0["a"], expect) and np.array_equal(r1["a"], expect)
        assert np.array_equal(r0["b"], [1.0, 1.0]) and np.array_equal(r1["b"], [1.0, 1.0])   # slot-gather idiom
        assert int(r0["calls"]) == 2


def _select_by_three_collectives(shards, k, cap):
    """Host model of the multi-rank order statistic in ba_select.h / mcp_ba::select_kth: two all-reduced 11-bit histogram
    passes over the IEEE-754 patterns of |x|, a (ranks x cap) slot table whose sum over the ranks gathers the candidates that
    share the 22-bit prefix, a local finish on the gathered table; beyond `cap` candidates the remaining passes run.
    Returns (value, number of all-reduces)."""
    keys = [np.abs(s).astype(np.float64).view(np.uint64) for s in shards]
    shifts, nbits = [53, 42, 31, 20, 9, 0], [11, 11, 11, 11, 11, 9]
    prefix, himask, n_ar = np.uint64(0), np.uint64(0), 0

    def one_pass(p, prefix, himask, k):
        hist = np.zeros(1 << nbits[p])
        for kk in keys:                                   # every rank's histogram, summed by the all-reduce
            sel = kk[(kk & himask) == prefix]
            np.add.at(hist, ((sel >> np.uint64(shifts[p])) & np.uint64((1 << nbits[p]) - 1)).astype(np.int64), 1.0)
        cum = np.cumsum(hist)
        b = int(np.searchsorted(cum, k, side="right"))
        b = min(b, len(hist) - 1)
        k_in = k - (int(cum[b - 1]) if b > 0 else 0)
        return b, k_in, int(hist[b])

    for p in range(2):
        b, k, count = one_pass(p, prefix, himask, k)
        n_ar += 1
        prefix |= np.uint64(b) << np.uint64(shifts[p])
        himask = ~np.uint64(0) << np.uint64(shifts[p])
    world = len(shards)
    table, counts = np.zeros((world, cap)), np.zeros(world)
    for r, kk in enumerate(keys):                          # each rank fills only its own slot; the sum is the gather
        mine = kk[(kk & himask) == prefix].view(np.float64)[:cap]
        table[r, :len(mine)] = mine
        counts[r] = len(mine)
    n_ar += 1
    if count <= cap:                                       # the same decision on every rank: `count` is all-reduced
        cand = np.concatenate([table[r, :int(counts[r])] for r in range(world)])
        assert len(cand) == count
        return float(np.sort(cand)[k]), n_ar
    for p in range(2, 6):
        b, k, _ = one_pass(p, prefix, himask, k)
        n_ar += 1
        prefix |= np.uint64(b) << np.uint64(shifts[p])
        himask = ~np.uint64(0) << np.uint64(shifts[p])
    return float(np.array([prefix], dtype=np.uint64).view(np.float64)[0]), n_ar


def test_three_collective_order_statistic_is_exact_on_shards():
    """The exchange pattern of the sharded median modelled on the host: exact for ragged shards, empty ranks,
    heavy ties, all-zero data (prefix 0: the zero padding of the slot table must not be counted -- hence the per-rank counts)
    and table overflow (falls back to the histogram passes)."""
    rng = np.random.default_rng(8)
    cases = []
    cases.append([rng.gamma(2.0, size=n) for n in (1000, 37, 0, 512)])
    cases.append([np.round(rng.gamma(2.0, size=n), 1) for n in (300, 300)])               # many exact ties
    cases.append([np.zeros(50), np.zeros(7)])                                             # everything shares prefix 0
    cases.append([np.concatenate([np.zeros(40), rng.gamma(2.0, size=5)]), rng.gamma(2.0, size=3)])
    cases.append([np.full(200, 3.25), np.full(100, 3.25), rng.gamma(2.0, size=10)])       # more equal values than a slot holds
    cases.append([-rng.gamma(2.0, size=64), rng.gamma(2.0, size=64)])                     # |x| is what is ranked
    for shards in cases:
        allv = np.sort(np.abs(np.concatenate(shards)))
        for k in sorted({0, len(allv)//2, len(allv) - 1}):
            for cap in (4096, 16):
                got, n_ar = _select_by_three_collectives(shards, k, cap)
                assert got == allv[k], (k, cap, got, allv[k])
                assert n_ar in (3, 7)
    got, n_ar = _select_by_three_collectives(cases[4], 150, 16)
    assert n_ar == 7 and got == 3.25                                                      # the overflow path was taken
    got, n_ar = _select_by_three_collectives(cases[0], 700, 4096)
    assert n_ar == 3
