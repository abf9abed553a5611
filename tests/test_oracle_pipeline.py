"""CPU tests of the oracle itself (no GPU): known-answer tracking against the generating trajectory of the
analytic scene, structural properties of every stage, and the quirks that the restatement must keep."""
import os

import numpy as np
import pytest

import efo
from conftest import rgba_of

FX, FY, CX, CY = 528.0, 528.0, 320.0, 240.0


def pose_err(T, Tr):
    dt = float(np.linalg.norm(T[:3, 3] - Tr[:3, 3]))
    dR = T[:3, :3].T @ Tr[:3, :3]
    return dt, float(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))


def test_known_answer_tracking(oracle_state, frames):
    """Recovered pose vs the pose that generated the frames: independent of any restatement error."""
    dt, da = pose_err(oracle_state.pose(), frames[2][2])
    step = float(np.linalg.norm(frames[2][2][:3, 3]))
    assert step > 3e-3
    assert dt < 1.5e-3 and da < 1.5e-3, (dt, da)
    st = oracle_state.stats()
    assert st[1] > 200000 and st[3] > 10000 and st[5] > 15000  # ICP / RGB / SO3 support


def test_icp_normal_equations_properties(oracle_state):
    odo = oracle_state.odometry()
    T = oracle_state.pose()
    R = T[:3, :3].astype(np.float32)
    t = T[:3, 3].astype(np.float32)
    Rinv = np.linalg.inv(R).astype(np.float32)
    args = lambda Rc, tc: (Rc, tc, odo.buffer("vmap_curr", 1), odo.buffer("nmap_curr", 1), Rinv, t, (FX / 2, FY / 2, CX / 2, CY / 2),
                           odo.buffer("vmap_g_prev", 1), odo.buffer("nmap_g_prev", 1), 0.10, float(np.sin(20 * 3.14159254 / 180)))
    A, b, res = efo.icp_step(*args(R, t))
    assert np.array_equal(A, A.T)
    assert np.all(np.linalg.eigvalsh(A.astype(np.float64)) > -1e-3 * np.abs(A).max())
    assert 0 < res[1] <= 320 * 240
    # at the converged pose the gradient is tiny compared with one of a displaced pose
    A2, b2, res2 = efo.icp_step(*args(R, t + np.array([0.01, 0, 0], np.float32)))
    assert np.linalg.norm(b) < 0.5 * np.linalg.norm(b2)  # (the joint ICP+RGB optimum is not exactly ICP's)
    assert res2[0] / res2[1] > res[0] / res[1]


def test_quirk_q1_next_depth_equals_last_depth(oracle_state):
    odo = oracle_state.odometry()
    for l in range(3):
        a, b = odo.buffer("lastDepth", l), odo.buffer("nextDepth", l)
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def test_quirk_q3_only_x_plane_gets_nan():
    depth = np.zeros((6, 8), np.uint16)
    depth[1:4, 2:6] = 1000
    init = np.full((18, 8), 5.0, np.float32)
    v = efo.create_vmap(depth, 100.0, 100.0, 4.0, 3.0, 20.0, vmap=init.copy())
    bad = np.isnan(v[:6])
    assert bad.sum() == 48 - 12
    assert np.all(v[6:12][bad] == 5.0) and np.all(v[12:][bad] == 5.0)


def test_quirk_q6_sobel_border_and_q7_pyrdown_window():
    img = (np.arange(12 * 16).reshape(12, 16) % 251 + 1).astype(np.uint8)
    dx, dy = efo.derivative_images(img)
    # interior: plain 3x3 correlation with the kernel indexed 8..0 == convolution
    gx = np.array([[0.52201, 0, -0.52201], [0.79451, 0, -0.79451], [0.52201, 0, -0.52201]], np.float32)
    y, x = 5, 7
    acc = np.float32(0)
    k = 8
    for j in range(y - 1, y + 2):
        for i in range(x - 1, x + 2):
            acc += np.float32(img[j, i]) * gx.reshape(9)[k]
            k -= 1
    assert dx[y, x] == np.int16(int(acc))
    # Q7: the 5x5 window never reads the last row/col => constant image keeps its value, a spike in the last column is ignored
    f = np.full((12, 16), 2.0, np.float32)
    f[:, 15] = 100.0
    d = efo.pyr_down_gauss_f(f)
    assert np.all(d == 2.0)


def test_rgb_residual_is_int_and_asymmetric_window(oracle_state):
    odo = oracle_state.odometry()
    l = 1
    K = np.array([[FX / 2, 0, CX / 2], [0, FY / 2, CY / 2], [0, 0, 1]])
    corres, sigma, count = efo.rgb_residual(3 * 3 * 64.0, odo.buffer("dIdx", l), odo.buffer("dIdy", l), odo.buffer("lastDepth", l),
                                            odo.buffer("nextDepth", l), odo.buffer("lastImage", l), odo.buffer("nextImage", l), 0.07,
                                            np.zeros(3, np.float32), np.eye(3, dtype=np.float32))
    v = corres["valid"] != 0
    assert count == v.sum() > 1000
    assert sigma == int((corres["diff"][v].astype(np.int64) ** 2).sum())
    assert not v[:, -5:].any() and not v[-1, :].any()          # j0 < cols-5, i < rows-1
    ys, xs = np.nonzero(v)
    assert np.array_equal(corres["one"][v][:, 0], xs) and np.array_equal(corres["one"][v][:, 1], ys)
    assert np.array_equal(corres["zero"][v], corres["one"][v])  # identity warp


def test_preprocessing_and_seeding_properties(frames):
    rgb, depth, _ = frames[0]
    f = efo.filter_depth(depth, 3.0)
    assert np.array_equal(f == 0, (depth > 3000) | (depth < 300))
    assert np.abs(f.astype(int) - depth.astype(int))[f > 0].max() <= 60   # sigma_colour = 30 mm
    assert np.median(np.abs(f.astype(int) - depth.astype(int))[f > 0]) <= 2
    cam = efo.make_cam(640, 480, FX, FY, CX, CY)
    dm, dmf = efo.metricise_depth(depth, 3.0), efo.metricise_depth(f, 3.0)
    s = efo.seed_map(cam, rgb, dm, dmf, 1, 20.0)
    assert len(s) == (dm > 0).sum()
    assert np.all(s[:, 6] == 1) and np.all(s[:, 7] == 1) and np.all(s[:, 5] == 0)
    assert np.all((s[:, 3] > 0) & (s[:, 3] <= 1))
    nrm = np.linalg.norm(s[:, 8:11], axis=1)
    assert np.nanmax(np.abs(nrm - 1)) < 1e-5
    # column-major emission order (FeedbackBuffer.cpp:44-52): x of the first surfels is constant, y increases
    assert s[0, 0] == s[1, 0] and s[1, 1] > s[0, 1]
    # colour is the packed 24-bit integer of the source pixel
    assert s[0, 4] == float((int(rgb[0, 0, 0]) << 16) + (int(rgb[0, 0, 1]) << 8) + int(rgb[0, 0, 2]))


def test_index_map_projects_to_its_pixel(oracle_state):
    idx = oracle_state.buffer("index")
    vc = oracle_state.buffer("vertConf")
    ys, xs = np.nonzero(idx)
    assert len(ys) > 100000
    u = np.float32(FX) * vc[ys, xs, 0] / vc[ys, xs, 2] + np.float32(CX)
    v = np.float32(FY) * vc[ys, xs, 1] / vc[ys, xs, 2] + np.float32(CY)
    # index_map.vert hands the point over in NDC and the viewport transform brings it back (restated since round 5, oracle/efo_map.cpp): the
    # window position is floor(((ndc + 1) / 2) * size) on the float NDC value — the pixel of floor(u) except for a point within ~1e-5 px of an edge
    f32 = np.float32
    xn = ((u - f32(320)) / f32(320)).astype(f32)
    yn = ((v - f32(240)) / f32(240)).astype(f32)
    assert np.array_equal(np.floor((xn.astype(np.float64) + 1.0) * 0.5 * 640).astype(int), xs)
    assert np.array_equal(np.floor((yn.astype(np.float64) + 1.0) * 0.5 * 480).astype(int), ys)
    off = (np.floor(u).astype(int) != xs) | (np.floor(v).astype(int) != ys)
    assert off.mean() < 1e-3
    assert np.all(np.minimum(np.abs(u[off] - np.rint(u[off])), np.abs(v[off] - np.rint(v[off]))) < 1e-3)


def test_map_order_and_monotone_times(oracle_state):
    m = oracle_state.map()
    assert np.all(np.diff(m[:, 6]) >= 0)          # ordered by init time: Deformation.cpp:294-296 relies on it
    assert m[:, 7].max() == oracle_state.tick() - 1
    assert np.all(m[:, 7] >= m[:, 6])


def test_clean_is_idempotent_on_a_static_view(oracle_state):
    cam = efo.make_cam(640, 480, FX, FY, CX, CY)
    T = oracle_state.pose()
    tick = oracle_state.tick()
    m = oracle_state.map()
    TD = 2147483647 // 2
    idx, vc, ct, nr = efo.predict_indices(cam, T, tick, m, 20.0, TD)
    a = efo.clean(cam, T, tick, idx, vc, ct, nr, 10.0, TD, 20.0, m, np.zeros((0, 12), np.float32))
    idx, vc, ct, nr = efo.predict_indices(cam, T, tick, a, 20.0, TD)
    b = efo.clean(cam, T, tick, idx, vc, ct, nr, 10.0, TD, 20.0, a, np.zeros((0, 12), np.float32))
    assert len(b) == len(a) and np.array_equal(a, b)


def test_fill_in_passthrough_equals_raw_frame(oracle_state, frames):
    cam = efo.make_cam(640, 480, FX, FY, CX, CY)
    rgb = frames[2][0]
    img = oracle_state.buffer("image")
    fi, fv, fn = efo.fill_in(cam, img, oracle_state.buffer("vertex"), oracle_state.buffer("normal"),
                             oracle_state.buffer("depthFiltered"), rgb, 1, 1)
    assert np.array_equal(fi[..., :3], rgb) and np.all(fi[..., 3] == 255)
    z = oracle_state.buffer("depthFiltered").astype(np.float32) / np.float32(1000.0)
    assert np.array_equal(fv[..., 2], z)


def test_oracle_results_do_not_depend_on_thread_count(seq):
    """bench.py's "all cores" CPU baseline runs the same oracle with its parallel loops (bilateral rows, reduction blocks)
    split over std::threads: poses and maps must be bit-identical to the single-threaded run."""
    import efo
    out = []
    try:
        for n in (1, 5):
            efo.set_threads(n)
            f = efo.Fusion()
            for k in range(3):
                rgb, depth, _ = seq.frame(k)
                f.process_frame(rgb, depth, k)
            out.append((f.pose().copy(), f.map().copy(), np.asarray(f.stats(), np.float32)))
    finally:
        efo.set_threads(1)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    assert np.array_equal(out[0][2].view(np.uint32), out[1][2].view(np.uint32))


def test_bootstrap_drift_is_frame_to_frame_odometry():
    """Known answer + attribution: on noise-free synthetic input the recovered pose sits 5-6 mm from the
    generating trajectory after 16 frames (8-23 mm later on).  It is the ALGORITHM's bootstrap, not an implementation error: with
    the default confidence threshold (10) no surfel is stable for the first ~25 frames, the model prediction is empty, denseEnough
    (ElasticFusion.cpp:256-268,304-305) is false and the tracker is fed the fill-in maps — the previous frame's own depth — i.e. plain
    frame-to-frame odometry, which accumulates ~0.3 mm per frame here (the photometric term on the point-sampled procedural texture
    contributes about half: ICP only halves it).  With surfels stable at once (confidence 1) the tracker runs frame-to-MODEL from the
    second frame on and the error stays at the 1-2 mm of the depth quantisation.  Measured over 60 frames (round 2): default 22.8 mm /
    18 mrad max, no SO(3) 22.6, frame-to-frame RGB 23.4, pixel-centre rays (quirk Q4 removed) 25.6, half speed 24.1, ICP only 10.8,
    confidence 1: 2.7 mm / 1.5 mrad."""
    from elasticfusion_amd import synth
    seq = synth.Sequence(0xEF0002)
    efo.set_threads(min(os.cpu_count() or 1, 8))
    runs = {}
    for name, kw in (("default", {}), ("stable_at_once", dict(confidence=1.0))):
        o = efo.Fusion(**kw)
        for k in range(16):
            rgb, depth, T = seq.frame(k)
            o.process_frame(rgb, depth, k * 33333)
        runs[name] = float(np.linalg.norm(o.pose()[:3, 3] - T[:3, 3]))
        del o
    efo.set_threads(1)
    assert 0.003 < runs["default"] < 0.008, runs
    assert runs["stable_at_once"] < 0.0025 < runs["default"], runs


def test_oracle_resumed_from_a_checkpoint_continues_bit_for_bit():
    """efo_fusion_restore (the oracle's side of ef_map_upload + ef_restore_state; what the GPU parity test of BASELINE configs[2]'s pre-seeded
    map and the one-frame harness start from): a fresh oracle restored from {map, tick, pose as held, last frame} runs the next frames exactly
    like the oracle the checkpoint was taken from."""
    from elasticfusion_amd import synth
    sq = synth.Sequence(seed=0xEF0005, width=320, height=240)
    kw = dict(width=320, height=240, fx=sq.fx, fy=sq.fy, cx=sq.cx, cy=sq.cy, maxSurfels=1 << 19, confidence=2.0)
    frames = [sq.frame(k) for k in range(9)]
    a = efo.Fusion(**kw)
    for k in range(6):
        a.process_frame(frames[k][0], frames[k][1], k * 33333)
    ck = a.checkpoint(frames[5][0], frames[5][1])
    b = efo.Fusion(**kw)
    b.restore(ck)
    assert b.tick() == a.tick() and b.map_count() == a.map_count() and np.array_equal(b.pose(), a.pose())
    for k in range(6, 9):
        for o in (a, b):
            o.process_frame(frames[k][0], frames[k][1], k * 33333)
        assert np.array_equal(np.asarray(a.stats(), np.float32).view(np.uint32), np.asarray(b.stats(), np.float32).view(np.uint32)), k
        assert np.array_equal(a.pose(), b.pose()) and a.map_count() == b.map_count(), k
    assert np.array_equal(a.map().view(np.uint32), b.map().view(np.uint32))
