"""synth.partition: one map -> per-rank shards; host logic, no GPU."""
import numpy as np
import pytest

from mcptam_amd import synth


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_partition_blocks_are_contiguous_in_source_mkf_and_balanced(world):
    p = synth.make_config("c2", n_mkf=24, n_points=4000)
    shards = synth.partition(p, world)
    assert len(shards) == world
    pts = np.concatenate([s.part["points"] for s in shards])
    ms = np.concatenate([s.part["meas"] for s in shards])
    assert np.array_equal(np.sort(pts), np.arange(p.n_points)) and np.array_equal(np.sort(ms), np.arange(p.n_meas))
    # contiguous blocks after sorting by source MKF: the source-MKF ranges of consecutive shards do not interleave
    hi = -1
    for s in shards:
        if s.n_points:
            assert s.pt_src[:, 0].min() >= hi and (np.diff(s.pt_src[:, 0]) >= 0).all()
            hi = s.pt_src[:, 0].max()
    # every measurement lives with its point, balanced to within one point's measurements
    counts = [s.n_meas for s in shards]
    assert max(counts) - min(counts) <= 2 * 8
    for s in shards:
        assert np.array_equal(p.ms_pt[s.part["meas"]], s.part["points"][s.ms_pt])
        assert np.array_equal(p.ms_uv[s.part["meas"]], s.ms_uv) and np.array_equal(p.pt_x[s.part["points"]], s.pt_x)
        assert (np.diff(s.part["meas"]) > 0).all()                     # the map's measurement order (MKF, camera, point) is kept
        assert np.array_equal(s.base_R, p.base_R) and np.array_equal(s.base_fixed, p.base_fixed)
    # the shards put together are the map again (up to the point order)
    m = synth.merge_shards(shards)
    assert m.n_points == p.n_points and m.n_meas == p.n_meas
    key = lambda q: sorted(zip(q.ms_mkf.tolist(), q.ms_cam.tolist(), map(tuple, np.round(q.ms_uv, 9).tolist())))
    assert key(m) == key(p)


def test_partition_of_a_replayed_map_dump(tmp_path):
    """A map that went through the DumpToFile text format (map_io) partitions like a synthetic one."""
    from mcptam_amd import map_io
    p = synth.make_config("tiny")
    m = map_io.map_from_problem(p)
    path = str(tmp_path / "map.dump")
    map_io.dump_map(path, m, precision=17)
    q = map_io.problem_from_map(map_io.load_map(path), {n: p.cams[i] for i, n in enumerate(m.cam_names)})
    shards = synth.partition(q, 2)
    assert sum(s.n_points for s in shards) == q.n_points and sum(s.n_meas for s in shards) == q.n_meas
    assert abs(shards[0].n_meas - shards[1].n_meas) <= 8


def test_single_rank_selection_matches_the_list():
    p = synth.make_config("tiny")
    all_ = synth.partition(p, 3)
    for r in range(3):
        one = synth.partition(p, 3, r)
        assert np.array_equal(one.part["points"], all_[r].part["points"]) and np.array_equal(one.ms_uv, all_[r].ms_uv)
