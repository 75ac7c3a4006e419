"""scripts/crosscheck_dump.py on dump pairs the oracle itself made (no GPU): a pair made with the default conventions is reproduced,
a pair made under a switched [3P-memory] convention is NOT -- and the script names the switch that explains it."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

from mcptam_amd import map_io, synth          # noqa: E402


def _make_pair(tmp_path, switches, iters, tag):
    import crosscheck_dump as cc
    from oracle import OracleBundle
    p = synth.make_config("c2", n_mkf=10, n_points=400)
    names = ["camera%d" % (c + 1) for c in range(len(p.cams))]
    before, after, cams = (str(tmp_path / ("%s_%s.dump" % (tag, k))) for k in ("before", "after", "cameras"))
    map_io.dump_map(before, map_io.map_from_problem(p, names), precision=17)
    map_io.dump_cameras(cams, dict(zip(names, p.cams)), precision=17)
    # the reference's side of the experiment, played by the oracle: replay the dump (as a real MCPTAM would hold it), adjust, dump again
    q = map_io.problem_from_map(map_io.load_map(before), map_io.load_cameras(cams))
    o = OracleBundle(q.cams, True, True, False)
    for k, v in switches:
        o.SetVariant(k, v)
    R, t, world, rc = cc.adjusted_state(o, q, iters)
    assert rc == iters
    map_io.dump_map(after, map_io.map_from_problem(q, names, state=(R, t, world)), precision=17)
    return before, after, cams


def test_a_pair_made_with_the_default_conventions_is_reproduced(tmp_path):
    import crosscheck_dump as cc
    before, after, cams = _make_pair(tmp_path, [], 6, "default")
    rep = cc.crosscheck(before, after, cams, iters=6, use_gpu=False)
    assert rep["ok"] and rep["default_oracle_reproduces_the_reference"]
    assert rep["candidates"][0]["within_tolerance"] and max(rep["candidates"][0][k] for k in ("pose_R", "pose_t", "points")) < 1e-9
    assert rep["reference_adjustment_moved"]["pose_t"] > 1e-4           # the adjustment did something, so the comparison means something
    # a wrong iteration count is visible too
    assert not cc.crosscheck(before, after, cams, iters=2, use_gpu=False)["ok"]


def test_a_pair_made_under_another_rejection_rule_is_flagged_and_explained(tmp_path):
    import crosscheck_dump as cc
    before, after, cams = _make_pair(tmp_path, [(0, 1e-3)], 6, "tau")
    rep = cc.crosscheck(before, after, cams, iters=6, use_gpu=False)
    assert not rep["ok"] and not rep["default_oracle_reproduces_the_reference"]
    assert rep["best_explaining_oracle_variant"] == "oracle: initial lambda tau = 1e-3"
    by = {c["candidate"]: c for c in rep["candidates"]}
    assert by["oracle: initial lambda tau = 1e-3"]["within_tolerance"]
    assert "initial lambda tau = 1e-3" in rep["verdict"]


def test_dumps_with_different_populations_are_refused(tmp_path):
    import crosscheck_dump as cc
    before, after, cams = _make_pair(tmp_path, [], 2, "a")
    p = synth.make_config("c2", n_mkf=10, n_points=300)
    other = str(tmp_path / "other.dump")
    map_io.dump_map(other, map_io.map_from_problem(p), precision=17)
    with pytest.raises(SystemExit):
        cc.crosscheck(before, other, cams, iters=2, use_gpu=False)
