"""GPU parity of the local loop closure's front half (ElasticFusion.cpp:447-527): every frame of the
synthetic revisit, the INACTIVE prediction, the model-to-model tracker's statistics and pose, the covariance gate, the sampled
surface constraints and — with a solver registered — the pose replacement and the deformed map must equal the oracle's."""
import numpy as np
import pytest

import efo
import loopscene

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64 if a.dtype == np.float64 else a.dtype)


def same_floats(a, b):
    """bit-identical, any NaN equal to any NaN"""
    a, b = np.asarray(a), np.asarray(b)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(bits(a[~na]), bits(b[~nb]))


def run_pair(solver_factory, builtin=False):
    from elasticfusion_amd import api
    efo.set_threads(8)
    ef = api.ElasticFusion(timeDelta=loopscene.TIME_DELTA, confidence=loopscene.CONFIDENCE, closeLoops=True, maxSurfels=1 << 21)
    o = efo.Fusion(timeDelta=loopscene.TIME_DELTA, confidence=loopscene.CONFIDENCE, maxSurfels=1 << 21)
    o.set_close_loops(True)
    sh = so = None
    if solver_factory:
        sh, so = solver_factory(), solver_factory()
        ef.setLoopSolver(sh)
        o.set_loop_solver(so)
    if builtin:
        # the engine's built-in optimiser on one side; on the other the oracle's frame loop calling the SAME optimiser (its host entry
        # point) on the graph sampled from the oracle's map, with Deformation::lastDeformTime kept here
        ef.useBuiltinLoopSolver(True)
        state = dict(last=0)

        def oracle_solver(info, cons):
            r = api.solve_local_deformation(efo.sample_graph(o.map()), cons, o.tick(), state["last"])
            if r is None:
                return None
            state["last"] = o.tick()
            return r[0]
        o.set_loop_solver(oracle_solver)
    opened = applied = 0
    for i, (rgb, depth, T) in enumerate(loopscene.frames()):
        ef.processFrame(rgb, depth, i * 33333, in_T_wc=T)
        o.process_frame(rgb, depth, i * 33333, T_wc=T)
        a, ca = ef.localLoop()
        b, cb = o.local_loop()
        for f in ("attempted", "cov_ok", "gates_ok", "n_constraints", "applied", "graph_nodes"):
            assert getattr(a, f) == getattr(b, f), (i, f, getattr(a, f), getattr(b, f))
        assert same_floats(np.array(a.stats, np.float32), np.array(b.stats, np.float32)), (i, list(a.stats), list(b.stats))
        assert same_floats(np.array(a.cov_diag), np.array(b.cov_diag)), (i, list(a.cov_diag), list(b.cov_diag))
        assert same_floats(np.array(a.T_wc_curr), np.array(b.T_wc_curr)), i
        assert same_floats(np.array(a.T_wc_est), np.array(b.T_wc_est)), (i, np.array(a.T_wc_est) - np.array(b.T_wc_est))
        assert np.array_equal(bits(ca), bits(cb)), i
        if i > 0:
            assert np.array_equal(ef.image("old_time"), o.old_buffer("time")), i
            assert np.array_equal(ef.image("old_image"), o.old_buffer("image")), i
            assert same_floats(ef.image("old_vertex"), o.old_buffer("vertex")), i
            assert same_floats(ef.image("old_normal"), o.old_buffer("normal")), i
        # identical as floats; the double matrices may differ in the last bit of a small element (two libms' sin / cos)
        assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32)) and np.abs(ef.get_T_wc() - o.pose()).max() < 1e-15, i
        assert ef.lastCount() == o.map_count(), i
        if i % 5 == 4:   # Resize::{image,vertex} at the fern database's factor 8 and the constraint grid's factor 20
            for f in (8, 20):
                assert np.array_equal(ef.imageResized("image", f), efo.resize_nearest(o.buffer("image"), f)), (i, f)
                assert same_floats(ef.imageResized("vertex", f), efo.resize_nearest(o.buffer("vertex"), f)), (i, f)
                assert np.array_equal(ef.imageResized("old_time", f), efo.resize_nearest(o.old_buffer("time"), f)), (i, f)
        if i % 5 == 4:   # Deformation::sampleGraphModel of the map as it stands (ElasticFusion.cpp:593)
            nodes = ef.sampleGraph()
            assert np.array_equal(bits(nodes), bits(efo.sample_graph(o.map()))) and len(nodes) == (o.map_count() - 1) // 5000 + 1
            assert (np.diff(nodes[:, 3]) >= 0).all()    # the assumption Deformation.cpp:295-297 asserts
        opened += a.gates_ok
        applied += a.applied
    assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32))
    ef.close()
    efo.set_threads(1)
    return opened, applied, sh, so


def test_front_half_matches_oracle_frame_by_frame():
    opened, applied, _, _ = run_pair(None)
    assert opened >= 4 and applied == 0


def test_accepted_deformation_matches_oracle():
    opened, applied, sh, so = run_pair(loopscene.OneShotSolver)
    assert opened >= 2 and applied == 1
    assert sh.accepted == so.accepted == 1 and len(sh.calls) == len(so.calls)
    for (na, ca), (nb, cb) in zip(sh.calls, so.calls):
        assert na == nb and np.array_equal(bits(ca), bits(cb))


def test_builtin_solver_closes_local_loops_end_to_end():
    """closeLoops with the built-in deformation-graph optimiser: every time the gates open the map is deformed and the pose replaced,
    with no host code of the caller involved — frame by frame identical to the oracle's frame loop around the same optimiser; and
    the deformation does what it is for: the drift the scene builds in is largely gone from the registration after it."""
    opened, applied, _, _ = run_pair(None, builtin=True)
    assert opened >= 2 and applied == opened


def test_thresholds_and_state_errors():
    """countThresh / errThresh / covThresh of the constructor reach the gates; a solver can only be registered on a context
    created with closeLoops."""
    from elasticfusion_amd import api
    ef = api.ElasticFusion(timeDelta=loopscene.TIME_DELTA, confidence=loopscene.CONFIDENCE, closeLoops=True, countThresh=20000,
                           maxSurfels=1 << 21)
    fr = loopscene.frames()
    for i, (rgb, depth, T) in enumerate(fr[:16]):
        ef.processFrame(rgb, depth, i, in_T_wc=T)
    info, cons = ef.localLoop()
    assert info.attempted and info.stats[1] > 20000
    with pytest.raises(api.EFError):
        api.ElasticFusion().setLoopSolver(lambda i, c: None)      # open-loop context
    ef.close()
