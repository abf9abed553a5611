"""GPU parity, operator tier, map side: the HIP replacements of the reference's GLSL passes (bilateral /
metric depth, first-frame seeding, index splat, surfel splat, fill-in, denseEnough, fuse, clean) against the
CPU oracle on identical inputs from an oracle run over synthetic sequence 1.

Every pass is elementwise per pixel / per surfel with a specified operation order, so the bar is bit-exact
(floats compared by bit pattern, NaN == NaN), including surfel ORDER after the compactions.
"""
import numpy as np
import pytest

import efo
from test_gpu_ops_tracking import bits_equal

pytestmark = pytest.mark.gpu

W, H = 640, 480
FX, FY, CX, CY = 528.0, 528.0, 320.0, 240.0
MAXD = 20.0
TD = 2147483647 // 2


@pytest.fixture(scope="module")
def ops():
    from elasticfusion_amd import api
    return api.ops


@pytest.fixture(scope="module")
def cams():
    from elasticfusion_amd import api
    return api.ef_cam(W, H, FX, FY, CX, CY), efo.make_cam(W, H, FX, FY, CX, CY)


@pytest.fixture(scope="module")
def mature():
    """An oracle run with GT poses injected and a low confidence threshold, so that after a few frames the map
    has stable surfels, merged surfels, cleaned surfels and new unstable ones."""
    from elasticfusion_amd import synth
    seq = synth.Sequence(0xEF0002)
    f = efo.Fusion(confidence=1.0)
    fr = [seq.frame(k) for k in range(6)]
    for k in range(5):
        f.process_frame(fr[k][0], fr[k][1], k, T_wc=None if k == 0 else fr[k][2])
    return f, fr


def test_filter_and_metric_depth(ops, frames):
    raw = frames[1][1]
    f_r = efo.filter_depth(raw, 3.0)
    f = ops.filter_depth(raw, 3.0)
    assert np.array_equal(f, f_r)
    assert bits_equal(ops.metricise_depth(raw, 3.0), efo.metricise_depth(raw, 3.0))
    assert bits_equal(ops.metricise_depth(f, 3.0), efo.metricise_depth(f_r, 3.0))


def test_filter_depth_gates_and_borders(ops):
    rng = np.random.RandomState(11)
    raw = rng.randint(250, 3200, size=(48, 80)).astype(np.uint16)
    raw[rng.rand(*raw.shape) < 0.15] = 0
    raw[10:20, 10:30] = 1500 + rng.randint(-20, 20, size=(10, 20))
    assert np.array_equal(ops.filter_depth(raw, 3.0), efo.filter_depth(raw, 3.0))


def test_seed_map(ops, cams, frames):
    cam, ocam = cams
    rgb, depth, _ = frames[0]
    depth = depth.copy()
    depth[100:140, 200:260] = 0       # holes
    depth[300:310, 50:400] = 305      # near the 300 mm gate: the filtered value can fall below it
    depth[311:320, 50:400] = 296
    dm = efo.metricise_depth(depth, 3.0)
    dmf = efo.metricise_depth(efo.filter_depth(depth, 3.0), 3.0)
    s_r = efo.seed_map(ocam, rgb, dm, dmf, 1, MAXD)
    s = ops.seed_map(cam, rgb, dm, dmf, 1, MAXD)
    assert len(s) == len(s_r) > 200000
    assert bits_equal(s, s_r)


def test_predict_indices(ops, cams, mature):
    cam, ocam = cams
    f, fr = mature
    surf = f.map()
    T = fr[5][2]
    ref = efo.predict_indices(ocam, T, f.tick(), surf, MAXD, TD)
    got = ops.predict_indices(cam, T, f.tick(), surf, MAXD, TD)
    assert np.array_equal(got[0], ref[0]) and (ref[0] > 0).sum() > 100000
    for a, b in zip(got[1:], ref[1:]):
        assert bits_equal(a, b)


def test_inactive_prediction(ops, cams, mature):
    """IndexMap::INACTIVE (ElasticFusion.cpp:451-459): time = 0, maxTime = tick - timeDelta"""
    cam, ocam = cams
    f, fr = mature
    surf = f.map()
    T = fr[5][2]
    ref = efo.combined_predict(ocam, T, surf, MAXD, 1.0, 0, f.tick() - 3, 3)
    got = ops.combined_predict(cam, T, surf, MAXD, 1.0, 0, f.tick() - 3, 3)
    assert (ref[1][..., 2] > 0).sum() > 1000
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[3], ref[3])
    assert bits_equal(got[1], ref[1]) and bits_equal(got[2], ref[2])


def test_combined_predict_fill_dense(ops, cams, mature):
    cam, ocam = cams
    f, fr = mature
    surf = f.map()
    assert (surf[:, 3] >= 1.0).sum() > 50000, "fixture should hold stable surfels"
    T = fr[5][2]
    ref = efo.combined_predict(ocam, T, surf, MAXD, 1.0, f.tick(), f.tick(), TD)
    got = ops.combined_predict(cam, T, surf, MAXD, 1.0, f.tick(), f.tick(), TD)
    assert (ref[1][..., 2] > 0).sum() > 50000
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[3], ref[3])
    assert bits_equal(got[1], ref[1]) and bits_equal(got[2], ref[2])
    df = f.buffer("depthFiltered")
    rgb = fr[4][0]
    fr_ref = efo.fill_in(ocam, ref[0], ref[1], ref[2], df, rgb)
    fr_got = ops.fill_in(cam, got[0], got[1], got[2], df, rgb)
    assert np.array_equal(fr_got[0], fr_ref[0]) and bits_equal(fr_got[1], fr_ref[1]) and bits_equal(fr_got[2], fr_ref[2])
    pt_ref = efo.fill_in(ocam, ref[0], ref[1], ref[2], df, rgb, 1, 1)
    pt_got = ops.fill_in(cam, got[0], got[1], got[2], df, rgb, 1, 1)
    assert np.array_equal(pt_got[0], pt_ref[0]) and bits_equal(pt_got[1], pt_ref[1]) and bits_equal(pt_got[2], pt_ref[2])
    assert ops.dense_enough(cam, got[0]) == efo.dense_enough(ocam, ref[0])
    assert ops.dense_enough(cam, fr_got[0]) == efo.dense_enough(ocam, fr_ref[0]) == True  # noqa: E712


def test_fuse_and_clean(ops, cams, mature):
    cam, ocam = cams
    f, fr = mature
    surf = f.map()
    rgb, depth, T = fr[5]
    tick = f.tick()
    dm = efo.metricise_depth(depth, 3.0)
    dmf = efo.metricise_depth(efo.filter_depth(depth, 3.0), 3.0)
    idx, vc, ct, nr = efo.predict_indices(ocam, T, tick, surf, MAXD, TD)
    s_ref, nu_ref = efo.fuse(ocam, T, tick, rgb, dm, dmf, idx, vc, ct, nr, MAXD, 0.8, surf)
    s_got, nu_got = ops.fuse(cam, T, tick, rgb, dm, dmf, idx, vc, ct, nr, MAXD, 0.8, surf)
    assert (nu_ref[:, 7] == -1).sum() > 1000 and (nu_ref[:, 7] == -2).sum() > 10
    assert len(nu_got) == len(nu_ref) and bits_equal(nu_got, nu_ref)
    assert not np.array_equal(s_ref, surf)
    assert bits_equal(s_got, s_ref)
    idx2, vc2, ct2, nr2 = efo.predict_indices(ocam, T, tick, s_ref, MAXD, TD)
    out_ref = efo.clean(ocam, T, tick, idx2, vc2, ct2, nr2, 1.0, TD, MAXD, s_ref, nu_ref)
    out_got = ops.clean(cam, T, tick, idx2, vc2, ct2, nr2, 1.0, TD, MAXD, s_got, nu_got)
    assert len(out_ref) != len(s_ref) + len(nu_ref)
    assert len(out_got) == len(out_ref) and bits_equal(out_got, out_ref)


def test_clean_time_window(ops, cams, mature):
    """closed-loop style time window (timeDelta = 3): old surfels are force-kept, new ones skip the neighbourhood test."""
    cam, ocam = cams
    f, fr = mature
    surf = f.map()
    T = fr[5][2]
    tick = f.tick()
    idx, vc, ct, nr = efo.predict_indices(ocam, T, tick, surf, MAXD, 3)
    got_idx = ops.predict_indices(cam, T, tick, surf, MAXD, 3)
    assert np.array_equal(got_idx[0], idx)
    nu = surf[:500].copy()
    nu[:, 7] = -2
    nu[::3, 7] = -1
    out_ref = efo.clean(ocam, T, tick, idx, vc, ct, nr, 1.0, 3, MAXD, surf, nu)
    out_got = ops.clean(cam, T, tick, idx, vc, ct, nr, 1.0, 3, MAXD, surf, nu)
    assert len(out_got) == len(out_ref) and bits_equal(out_got, out_ref)


def test_clean_with_deformation_graph(ops, cams):
    """GlobalModel::clean with a deformation graph (copy_unstable.vert:128-322) + the synthesized depth it
    tests against (G6).  HIP vs oracle on the same inputs: every surfel bit-identical, and the graph really moved the map."""
    import mapops
    cam, ocam = cams
    inp = mapops.make_inputs(W, H)
    graph = mapops.make_graph(inp)
    T, tick = inp["T"].reshape(4, 4), int(inp["tick"].reshape(-1)[0])
    d_ref = efo.synthesize_depth(ocam, T, inp["surf"], MAXD, mapops.CONF, tick, tick - 2, 65535)
    d_got = ops.synthesize_depth(cam, T, inp["surf"], MAXD, mapops.CONF, tick, tick - 2, 65535)
    assert bits_equal(d_got, d_ref) and (d_ref > 0).sum() > 10000
    for fern in (0, 1):
        ref = mapops.run_deform(efo, inp, graph, cam=ocam, isFern=fern)
        got = mapops.run_deform(mapops.HipMapOps(__import__("elasticfusion_amd.api", fromlist=["api"])), inp, graph, cam=cam, isFern=fern)
        assert got.shape == ref.shape and bits_equal(got, ref), fern
    plain = efo.clean(ocam, T, tick, inp["idx2"], inp["vc2"], inp["ct2"], inp["nr2"], mapops.CONF, mapops.TD, MAXD, inp["s2"], inp["nu"])
    assert (np.linalg.norm(ref[:, :3] - plain[:, :3], axis=1) > 1e-3).mean() > 0.9


def test_frame_tier_deformation(cams):
    """ef_set_deformation: the graph handed over before a frame is applied by that frame's clean (synthesizeDepth first),
    exactly as the oracle's frame loop does; the maps stay bit-identical afterwards, also on the following frames."""
    import mapops
    from elasticfusion_amd import api, synth
    seq = synth.Sequence(0xEF0002)
    ef, o = api.ElasticFusion(confidence=1.0), efo.Fusion(confidence=1.0)
    for k in range(8):
        rgb, depth, Tgt = seq.frame(k)
        if k == 5:
            g = mapops.make_graph(dict(surf=o.map(), tick=np.int32(o.tick())), n_nodes=32)
            ef.setDeformation(g)
            o.set_deformation(g)
        ef.processFrame(rgb, depth, k, in_T_wc=None if k == 0 else Tgt)
        o.process_frame(rgb, depth, k, T_wc=None if k == 0 else Tgt)
        assert ef.lastCount() == o.map_count(), k
        if k >= 5:
            assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32)), k
    ef.close()
