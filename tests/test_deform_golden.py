"""ef_solve_deformation replaying the calls the REFERENCE's own compiled Deformation::constrain answered in
tests/golden/deform_reference.npz (tools/make_deform_golden.py): global closures accepted and rejected, relative constraints, a local
closure after an earlier deformation, poses carried along.  Needs neither /root/reference nor a GPU."""
import os

import numpy as np

from elasticfusion_amd import api, build

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "deform_reference.npz")


def test_constrain_answers_match_the_compiled_reference():
    build.build()
    g = np.load(GOLDEN)
    outcomes = []
    for i in range(int(g["n"])):
        p = f"c{i}_"
        tick, fm, last, ok = (int(x) for x in g[p + "par"])
        rows = [(r[0:3], r[3:6], int(r[6]), int(r[7]), int(r[8]), int(r[9])) for r in g[p + "rows"]]
        poses = np.concatenate([g[p + "fern"], g[p + "traj"]]) if fm else g[p + "fern"]
        times = np.concatenate([g[p + "fern_t"], g[p + "traj_t"]]) if fm else g[p + "fern_t"]
        got = api.solve_deformation(g[p + "nodes"], rows, bool(fm), last, poses, times)
        assert got["accepted"] == bool(ok)
        outcomes.append(bool(ok))
        if not ok:
            assert np.array_equal(got["poses"], poses)
            continue
        ref = g[p + "graph"]
        assert np.array_equal(got["graph"][:, [0, 1, 2, 15]], ref[:, [0, 1, 2, 15]])
        assert np.abs(got["graph"][:, 3:15] - ref[:, 3:15]).max() <= 2e-6
        want = np.concatenate([g[p + "fern_out"], g[p + "traj_out"]]) if fm else g[p + "fern_out"]
        assert np.abs(got["poses"] - want).max() < 1e-6
        rel = g[p + "rel"]
        assert len(got["new_relative"]) == (0 if fm else len(rel))
        for r, q in zip(got["new_relative"], rel):
            assert np.abs(np.array(r[0]) - q[0:3]).max() < 1e-6 and np.array_equal(np.array(r[1]), q[3:6]) and (r[2], r[3]) == (q[6], q[7])
    assert outcomes == [True, False, False, True]
