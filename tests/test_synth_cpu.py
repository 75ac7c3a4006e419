"""The synthetic map generator is what the golden fixtures, the bench line and its parity check are built on: a change
to it silently changes all of them.  Pinned here by the structure of the maps it makes (exact integers) and by float checksums."""
import numpy as np
import pytest

EXPECTED = {
    ("tiny", ()): (240, 60, 7080, 4338, 149, 93, 126596.83030287696, 236.76476117831876, 29.20431993066712),
    ("c1", ()): (3000, 500, 748500, 121506, 2059, 1643, 1704541.9274189947, 2841.6029855314664, 48.253496804816905),
    ("c2", ()): (80000, 10000, 399960000, 13852935, 56015, 196780, 44527814.544717446, 65373.69513871673, 249.15410967321887),
    ("calib", ()): (3600, 600, 1078200, 140554, 2559, 2388, 1955204.7627409026, 2818.955908894132, 57.78629110791331),
    ("metric", (("shard", 1),)): (400000, 50000, 9999800000, 279562730, 279913, 4765742, 223675476.05335647, 235466.24498566307, 999.1486108319405),
}


@pytest.mark.parametrize("key", sorted(EXPECTED, key=str), ids=lambda k: k[0] + ("-shard1" if k[1] else ""))
def test_generator_makes_the_pinned_maps(key):
    from mcptam_amd import synth
    name, kw = key
    p = synth.make_config(name, **dict(kw))
    got = (int(p.n_meas), int(p.n_points), int(np.asarray(p.ms_pt, dtype=np.int64).sum()),
           int((np.asarray(p.ms_mkf, dtype=np.int64)*7 + p.ms_cam).sum()), int(np.asarray(p.ms_level).sum()), int(np.asarray(p.pt_src, dtype=np.int64).sum()))
    want = EXPECTED[key]
    assert got == want[:6]
    for a, b in zip((float(np.asarray(p.ms_uv).sum()), float(np.asarray(p.pt_x).sum()), float(np.asarray(p.base_t).sum())), want[6:]):
        assert abs(a - b) <= 1e-9*abs(b)
    # measurements are ordered MKF-major, then camera, then point (the order BundleAdjusterMulti adds them in, src/BundleAdjusterMulti.cc:168-200)
    k = (np.asarray(p.ms_mkf, dtype=np.int64)*16 + p.ms_cam)*(1 << 32) + p.ms_pt
    assert (np.diff(k) > 0).all()


def test_shards_share_the_trajectory_and_differ_in_the_map():
    from mcptam_amd import synth
    a, b = synth.make_config("c2", shard=0), synth.make_config("c2", shard=1)
    assert np.array_equal(a.base_R, b.base_R) and np.array_equal(a.base_t, b.base_t) and np.array_equal(a.true_base_t, b.true_base_t)
    assert not np.array_equal(a.pt_x, b.pt_x) and a.n_meas == b.n_meas
