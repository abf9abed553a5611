"""Closed-loop mode in steady state, at the reference's DEFAULT thresholds, free-running (VERDICT r2 weak #2 / next #4a).

The other closed-loop tests run a handful of frames with injected poses, a lowered confidence threshold, setTick jumps and (for the
accepted global closure) relaxed gates.  This one runs the product exactly as `class ElasticFusion(closeLoops = true)` runs it — confidence
10, icpCountThresh 35000, icpErrThresh 5e-5, covThresh 1e-5, the fern database with its default thresholds, the built-in optimiser — over
194 tracked frames of a dwell / sweep / dwell / sweep-back trajectory (tests/loopscene.py::SweepSequence; only the time window is shortened,
to 25 frames, so that the sequence stays affordable for the CPU oracle), and holds every frame against the oracle's frame loop (itself equal
to the compiled ElasticFusion.cpp step for step): tracker statistics, pose, surfel count, keyframes, the global closure's verdict, the local
closure's second tracker (six statistics, covariance gate, count / error gates), its constraints, the optimiser's verdict — and at the end
the deformed trajectory and the whole map, bit for bit.  What happens on the way, asserted: surfels become stable, tens of thousands of them
go inactive on their own when the camera looks away for longer than the time window, the inactive view is re-observed, the model-to-model
registration runs on real data for a hundred frames, the gates open by themselves, a deformation is accepted and re-activates the old
surface."""
import os

import numpy as np
import pytest

import efo
import loopscene
from test_gpu_global import oracle_with_ferns
from test_gpu_loop import bits, same_floats

pytestmark = pytest.mark.gpu
SEED = 5


def test_closed_loop_steady_state_matches_oracle_at_default_thresholds():
    from elasticfusion_amd import api
    td = loopscene.SweepSequence.TIME_DELTA
    frames = loopscene.sweep_frames()
    efo.set_threads(min(os.cpu_count() or 1, 32))
    ef = api.ElasticFusion(closeLoops=True, timeDelta=td)
    ef.useBuiltinLoopSolver(True)
    ef.enableGlobalClosure(seed=SEED)
    o, calls = oracle_with_ferns(SEED, timeDelta=td)
    thr = ef.getConfidenceThreshold()
    assert thr == 10.0
    opened = applied = second_tracker_frames = 0
    max_inactive = prev_inactive = 0
    inactive_before = inactive_after = None
    for k, (rgb, depth, _) in enumerate(frames):
        ef.processFrame(rgb, depth, k * 33333)
        o.process_frame(rgb, depth, k * 33333)
        st = np.asarray(ef.trackingStats()[0], np.float32)
        if k > 0:
            assert same_floats(st, np.asarray(o.stats(), np.float32)), (k, st, o.stats())
        assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32)) and np.abs(ef.get_T_wc() - o.pose()).max() < 1e-15, k
        assert ef.lastCount() == o.map_count(), (k, ef.lastCount(), o.map_count())
        g, go = ef.globalLoop(), o.global_loop()
        assert (g.attempted, g.closest, g.n_constraints, g.accepted, g.graph_nodes) == (go.attempted, go.closest, go.n_constraints, go.accepted, go.graph_nodes), k
        a, ca = ef.localLoop()
        b, cb = o.local_loop()
        for f in ("attempted", "cov_ok", "gates_ok", "n_constraints", "applied", "graph_nodes"):
            assert getattr(a, f) == getattr(b, f), (k, f, getattr(a, f), getattr(b, f))
        assert same_floats(np.array(a.stats, np.float32), np.array(b.stats, np.float32)), (k, list(a.stats), list(b.stats))
        assert same_floats(np.array(a.cov_diag), np.array(b.cov_diag)), (k, list(a.cov_diag), list(b.cov_diag))
        assert same_floats(np.array(a.T_wc_est), np.array(b.T_wc_est)), k
        # the constraints are T * (x, y, z, 1) in DOUBLE with T = the double pose matrices, which may differ in their last bit between the
        # GPU's and the host's libm (sin / cos of the update step: DESIGN.md 2, the 1e-15 of the pose assertion above); times and pins exact
        assert ca.shape == cb.shape and np.array_equal(bits(ca[:, 6:]), bits(cb[:, 6:])) and (len(ca) == 0 or np.abs(ca - cb).max() <= 4e-15), k
        second_tracker_frames += int(a.attempted and a.stats[1] > 0)
        if a.gates_ok and not opened:
            inactive_before = prev_inactive
        opened += a.gates_ok
        applied += a.applied
        if k % 6 == 5 or a.applied:
            assert len(ef.getFerns()) == len(o.ferns()), k
            m = ef.downloadMap()
            prev_inactive = int(((m[:, 7] <= ef.getTick() - 1 - td) & (m[:, 3] > thr)).sum())
            max_inactive = max(max_inactive, prev_inactive)
            if a.applied and inactive_after is None:
                inactive_after = prev_inactive
    # the regime, on the way
    assert max_inactive > 50000, max_inactive                       # stable surfels went inactive on their own
    assert second_tracker_frames > 60, second_tracker_frames        # the model-to-model tracker ran on a non-empty inactive view
    assert opened >= 1 and applied >= 1, (opened, applied)          # the reference's own gates opened, a deformation was accepted
    assert inactive_after is not None and inactive_after < 0.5 * inactive_before, (inactive_before, inactive_after)   # ... and re-activated the old surface
    assert any(c[3] and not c[0] for c in calls)                    # the oracle's side took the same decision through the same optimiser
    # keyframes, deformed trajectory, map
    assert len(ef.getFerns()) == len(o.ferns()) > 30
    for i in range(0, len(ef.getFerns()), 7):
        fa, fb = ef.getFerns().frame(i), o.ferns().frame(i)
        assert np.array_equal(fa["codes"], fb["codes"]) and fa["srcTime"] == fb["srcTime"] and np.abs(fa["T_wc"] - fb["T_wc"]).max() <= 1e-12, i
    Ts, _ = ef.trajectory()
    assert len(Ts) == len(frames) and np.abs(Ts - o.trajectory()).max() <= 1e-12
    assert ef.closure().counts()["deforms"] == applied
    assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32))
    ef.close()
    efo.set_threads(1)
