"""The oracle's local loop closure front half (ElasticFusion.cpp:447-527) on the synthetic revisit of tests/loopscene.py:
known-answer properties, since no run of the reference exists to compare with."""
import numpy as np
import pytest

import efo
import loopscene


@pytest.fixture(scope="module")
def run():
    efo.set_threads(8)
    o = efo.Fusion(timeDelta=loopscene.TIME_DELTA, confidence=loopscene.CONFIDENCE, maxSurfels=1 << 21)
    o.set_close_loops(True)
    solver = loopscene.OneShotSolver()
    o.set_loop_solver(solver)
    log = []
    for i, (rgb, depth, T) in enumerate(loopscene.frames()):
        o.process_frame(rgb, depth, i * 33333, T_wc=T)
        info, cons = o.local_loop()
        log.append(dict(info=info, cons=cons, pose=o.pose(), old_px=int((o.old_buffer("vertex")[..., 2] > 0).sum()), T_in=T))
    efo.set_threads(1)
    return log, solver


def test_gates_follow_the_revisit(run):
    log, solver = run
    assert log[0]["info"].attempted == 0                      # first frame only seeds the map
    # nothing is inactive before the camera has looked away for longer than the time window
    for e in log[1:5]:
        assert e["info"].attempted == 1 and e["old_px"] == 0 and e["info"].gates_ok == 0 and e["info"].n_constraints == 0
        assert np.isnan(e["info"].stats[0]) and e["info"].stats[1] == 0   # no correspondences: sqrt(0) / 0
    # back at view A the inactive surface fills the view and the gates open once enough of it is re-observed
    opened = [i for i, e in enumerate(log) if e["info"].gates_ok]
    assert opened and opened[0] >= loopscene.DRIFT_FROM + 2
    for i in opened:
        info = log[i]["info"]
        assert info.cov_ok and info.stats[1] > 35000 and info.stats[0] < 5e-5
        assert max(info.cov_diag) <= 1e-5
        assert 100 < info.n_constraints <= 32 * 24


def test_registration_reduces_the_drift(run):
    """the constraint targets (surface under T_wc_est) lie closer to where the surface truly is than the sources (under the
    drifted T_wc_curr): the model-to-model registration pulls the re-mapped surface back onto the inactive one"""
    log, _ = run
    opened = [e for e in log if e["info"].gates_ok]
    for e in opened[:1]:   # the first attempt; it is accepted, and later ones see the deformed map
        M = np.array(e["info"].T_wc_curr).reshape(4, 4)
        assert np.allclose(M, e["T_in"], atol=1e-12) or e["info"].applied == 0
        c = e["cons"]
        truth = e["T_in"] @ np.linalg.inv(loopscene.drift(np.eye(4)))
        pc = np.linalg.inv(M) @ np.c_[c[:, :3], np.ones(len(c))].T
        pt = (truth @ pc).T[:, :3]
        before = np.linalg.norm(c[:, 0:3] - pt, axis=1).mean()
        after = np.linalg.norm(c[:, 3:6] - pt, axis=1).mean()
        # (0.915 in the reference rounding — the default oracle since round 5 —, 0.865 with the fast build's fused multiply-adds: the same
        # registration, two roundings; the sign of the effect is the property)
        assert 3.5e-3 < before < 5.5e-3 and after < 0.95 * before, (before, after)


def test_constraints_are_the_sampled_surface_under_both_poses(run):
    log, _ = run
    e = next(e for e in log if e["info"].gates_ok)
    M = np.array(e["info"].T_wc_curr).reshape(4, 4)
    E = np.array(e["info"].T_wc_est).reshape(4, 4)
    c = e["cons"]
    assert len(c) == e["info"].n_constraints
    src_cam = (np.linalg.inv(M) @ np.c_[c[:, 0:3], np.ones(len(c))].T).T
    assert np.allclose((E @ src_cam.T).T[:, :3], c[:, 3:6], atol=1e-9)
    assert (src_cam[:, 2] > 0.3).all() and (src_cam[:, 2] < 3.1).all()
    assert (c[:, 6] > 0).all() and (c[:, 6] <= len(loopscene.KS)).all()
    assert (c[:, 7] == 1).all()                                # no deformation applied before: constraints are pinned


def test_accepted_deformation_replaces_the_pose_and_unpins(run):
    log, solver = run
    assert solver.accepted == 1
    k = next(i for i, e in enumerate(log) if e["info"].applied)
    info = log[k]["info"]
    assert info.graph_nodes == 16
    # T_wc_curr = T_wc_est, :525 (the quaternion is copied; the two 4x4 print-outs come from two conversion paths, one ulp apart at most)
    assert np.allclose(log[k]["pose"], np.array(info.T_wc_est).reshape(4, 4), rtol=0, atol=1e-15)
    # the deformation pass re-stamps the old surfels it moved into view (ElasticFusion.cpp:558-585): they are ACTIVE again
    assert log[k + 1]["old_px"] < 0.5 * log[k]["old_px"]
    later = [e for e in log[k + 1:] if e["info"].n_constraints]
    assert later and all((e["cons"][:, 7] == 0).all() and not e["info"].applied for e in later)   # deforms > 0: no more pins
