"""GPU parity, frame tier: ef_process_frame (== ElasticFusion::processFrame) against the CPU oracle on the same
synthetic frames.  Bars from BASELINE.json's north_star: pose within 1e-4 m / 1e-4 rad, fused surfel
positions / normals / radii within 1e-5 relative.
"""
import os

import numpy as np
import pytest

import efo

pytestmark = pytest.mark.gpu


def pose_err(T, Tr):
    dt = float(np.linalg.norm(T[:3, 3] - Tr[:3, 3]))
    dR = T[:3, :3].T @ Tr[:3, :3]
    ang = float(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))
    return dt, ang


def compare_maps(m, mr, rel=1e-5):
    """surfel lists in stable order; returns the fraction of surfels within `rel` (positions/normals/radii/conf)."""
    assert m.shape == mr.shape, (m.shape, mr.shape)
    scale = np.maximum(np.abs(mr), 1e-3)
    cols = [0, 1, 2, 3, 8, 9, 10, 11]
    ok = (np.abs(m[:, cols] - mr[:, cols]) <= rel * np.maximum(scale[:, cols], np.linalg.norm(mr[:, :3], axis=1, keepdims=True))).all(axis=1)
    ok &= (m[:, 4] == mr[:, 4]) & (m[:, 6] == mr[:, 6]) & (m[:, 7] == mr[:, 7])
    return float(ok.mean())


@pytest.fixture(scope="module")
def hip():
    from elasticfusion_amd import api
    return api


def test_first_frame_matches_oracle(hip, frames):
    rgb, depth, _ = frames[0]
    ef = hip.ElasticFusion()
    ef.processFrame(rgb, depth, 0)
    o = efo.Fusion()
    o.process_frame(rgb, depth, 0)
    assert ef.lastCount() == o.map_count()
    assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32))
    assert np.array_equal(ef.image("depth_filtered"), o.buffer("depthFiltered"))
    for a, b in (("fill_vertex", "fill_vertex"), ("fill_normal", "fill_normal"), ("vertex", "vertex")):
        x, y = ef.image(a), o.buffer(b)
        assert np.array_equal(np.isnan(x), np.isnan(y))
        assert np.array_equal(x[~np.isnan(x)], y[~np.isnan(y)])
    assert np.array_equal(ef.image("fill_image"), o.buffer("fill_image"))
    assert ef.getTick() == o.tick() == 2
    ef.close()


def test_tracking_and_fusion_sequence(hip, seq):
    """Free-running tracking + fusion on both sides.  The HIP reductions reproduce the reference's fp32 summation
    tree, so nothing is left to drift: the tracker statistics, the pose and every surfel must be IDENTICAL to the
    oracle's frame after frame (the north_star bars — 1e-4 m / 1e-4 rad, 1e-5 relative — are met with zero error)."""
    n = 16
    ef = hip.ElasticFusion()
    o = efo.Fusion()
    for k in range(n):
        rgb, depth, Tgt = seq.frame(k)
        ef.processFrame(rgb, depth, k * 33333)
        o.process_frame(rgb, depth, k * 33333)
        st, A, b = ef.trackingStats()
        so = o.stats()
        if k > 0:
            assert np.array_equal(np.asarray(st, np.float32).view(np.uint32), np.asarray(so, np.float32).view(np.uint32)), (k, st, so)
        if k > 0:   # T18 getCovariance: the same LU inverse of the same lastA => the same bits
            assert np.array_equal(ef.getCovariance(), efo.covariance(o.odometry().stats()[1])), k
        dt, da = pose_err(ef.get_T_wc(), o.pose())
        assert dt <= 1e-4 and da <= 1e-4, (k, dt, da)
        # the double-precision pose may differ in its last bits (libm sin/cos/atan2, test_gpu_ops_linalg.py); what the
        # float pipeline consumes from it does not
        assert np.abs(ef.get_T_wc() - o.pose()).max() <= 1e-15, (k, np.abs(ef.get_T_wc() - o.pose()).max())
        assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32)), k
        assert ef.lastCount() == o.map_count(), k
    # and both stay close to the generating trajectory (known-answer guard on the oracle itself): 5.65 mm / 1.04 mrad after 16 frames at the
    # default confidence of 10 — frame-to-frame odometry while the map has no stable surfel yet (2.7 mm with confidence 1)
    dt, da = pose_err(ef.get_T_wc(), seq.pose(n - 1))
    assert dt < 0.007 and da < 0.002, (dt, da)
    m, mr = ef.downloadMap(), o.map()
    assert m.shape == mr.shape
    assert compare_maps(m, mr) == 1.0
    assert np.array_equal(m.view(np.uint32), mr.view(np.uint32))
    traj, ts = ef.trajectory()
    assert len(traj) == n and ts[3] == 3 * 33333
    ef.close()


def test_host_pointer_frames_ride_a_ring_and_equal_the_device_pointer_path(hip, seq):
    """ef_process_frame (the reference's processFrame signature: HOST pointers, ElasticFusion.cpp:270-280) stages every frame in a ring of three
    pinned pairs, uploads it on a copy stream and lets the frame script read the landing buffers in place (round 6).  Thirty frames enqueued back
    to back with no getter in between — the host runs ahead of the GPU until the ring is full and then waits for the oldest slot's consumption
    marker — from ONE caller buffer that is overwritten right after every call (the call must have copied it): trajectory and map bit-identical
    to the same frames handed over as device pointers (the path every oracle comparison of this file pins)."""
    n = 30
    frames = [seq.frame(k) for k in range(n)]
    ref = hip.ElasticFusion()
    dev = [(hip.DevBuf.from_array(r), hip.DevBuf.from_array(d)) for r, d, _ in frames]
    for k in range(n):
        ref.processFrameDevice(dev[k][0].p.value, dev[k][1].p.value, k * 33333)
    traj_ref, _ = ref.trajectory()
    map_ref = ref.downloadMap()
    ref.close()
    for overlap in (0, 1):
        ef = hip.ElasticFusion()
        ef.setInputOverlap(overlap)
        rgb_buf = np.empty_like(frames[0][0])
        depth_buf = np.empty_like(frames[0][1])
        for k in range(n):
            rgb_buf[...] = frames[k][0]
            depth_buf[...] = frames[k][1]
            ef.processFrame(rgb_buf, depth_buf, k * 33333)
            rgb_buf[...] = 0          # the caller's buffers are free as soon as the call returns
            depth_buf[...] = 0
        traj, _ = ef.trajectory()
        assert len(traj) == n
        for k in range(n):
            assert np.array_equal(np.asarray(traj[k]), np.asarray(traj_ref[k])), (overlap, k)
        assert np.array_equal(ef.downloadMap().view(np.uint32), map_ref.view(np.uint32)), overlap
        assert ef.trackerFallbacks() == 0
        ef.close()


def test_pipelined_frames_match_oracle(hip, seq):
    """Frames enqueued back to back with no getter in between: frame k+1's input stage (copy-in, bilateral filter,
    frame pyramids) runs on the second stream while frame k is still being fused.  The whole trajectory and the
    final map must still be the oracle's, bit for bit, and identical to a run with the overlap switched off."""
    n = 12
    o = efo.Fusion()
    for k in range(n):
        rgb, depth, _ = seq.frame(k)
        o.process_frame(rgb, depth, k * 33333)
    runs = []
    # 4: the input stream restricted to every 4th CU (ef_set_input_cu_mask); 2: the bilateral filter already DURING the previous tracker — which a
    # context whose tracker is the persistent launch (all 256 CUs to itself) runs as mode 1: no frame may end up on the one-workgroup fallback
    for overlap in (True, False, 4, 2):
        ef = hip.ElasticFusion()
        if overlap == 4:
            ef.setInputCuMask(4)
        ef.setInputOverlap(2 if overlap == 2 else int(bool(overlap)))
        for k in range(n):
            rgb, depth, _ = seq.frame(k)
            ef.processFrame(rgb, depth, k * 33333)
        traj, _ = ef.trajectory()
        runs.append((traj, ef.downloadMap(), ef.image("depth_filtered"), ef.image("fill_image")))
        assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32))
        assert ef.lastCount() == o.map_count()
        assert np.array_equal(runs[-1][1].view(np.uint32), o.map().view(np.uint32))
        assert np.array_equal(runs[-1][2], o.buffer("depthFiltered"))
        assert ef.trackerFallbacks() == 0, overlap
        ef.close()
    for other in (1, 2, 3):
        for a, b in zip(runs[0], runs[other]):
            assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))


@pytest.mark.parametrize("cfg", [dict(), dict(icpThresh=100.0), dict(so3=False), dict(fastOdom=True)], ids=["default", "icp_only", "no_so3", "fastOdom"])
def test_persistent_and_per_step_tracker_scripts_agree(hip, seq, cfg):
    """The whole tracker as ONE persistent launch of 256 workgroups (the default: k_track_ref in the reference order since round 5, k_track_fast in
    the opt-in fast build), as one launch per step (ef_set_persistent_tracker(ctx, 0), the round-2 script) and — reference-order builds —
    as round 3's launch of the small levels on 128 workgroups (mode 2: k_track_small) are the same arithmetic in the same order: statistics,
    pose, trajectory and map must be bit-identical, frame after frame — whichever of them the oracle comparisons of this file ran."""
    n = 14
    runs = []
    # fused: level-0 update step inside the search launch (ef_set_fused_step); resident: the persistent launch's level-resident pixel data (round 6,
    # ef_set_resident_levels; off = round 5's streaming launch)
    for persistent, fused, resident in ((1, False, True), (0, False, True), (1, True, True), (2, False, True), (1, False, False)):
        ef = hip.ElasticFusion(**cfg)
        ef.setPersistentTracker(persistent)
        ef.setFusedStep(fused)
        ef.setResidentLevels(resident)
        rec = []
        for k in range(n):
            rgb, depth, _ = seq.frame(k)
            ef.processFrame(rgb, depth, k * 33333)
            st, A, b = ef.trackingStats()
            rec.append((np.asarray(st, np.float32).view(np.uint32).copy(), A.copy(), b.copy(), ef.getPoseQT()))
        ef.synchronize()                      # also: no barrier of the persistent launches timed out
        runs.append((rec, ef.downloadMap()))
        ef.close()
    for other in (1, 2, 3, 4):
        for k in range(1, n):
            for x, y in zip(runs[0][0][k], runs[other][0][k]):
                assert np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y), (other, k, x, y)
        assert np.array_equal(runs[0][1].view(np.uint32), runs[other][1].view(np.uint32)), other


VARIANTS = {
    "fastOdom": (dict(fastOdom=True), dict(fastOdom=1), None),
    "no_so3": (dict(so3=False), dict(so3=0), None),
    "no_pyramid": (dict(), dict(pyramid=0), lambda ef: ef.setPyramid(False)),
    "icp_only": (dict(icpThresh=100.0), dict(icpWeight=100.0), None),
    "rgb_only": (dict(), dict(rgbOnly=1), lambda ef: ef.setRgbOnly(True)),
    "frame_to_frame_rgb": (dict(frameToFrameRGB=True), dict(frameToFrameRGB=1), None),
    "low_icp_weight": (dict(icpThresh=2.5), dict(icpWeight=2.5), None),
    "depth_cut_2m": (dict(depthCut=2.0), dict(depthCut=2.0), None),
    "low_confidence_time_window": (dict(confidence=2.0, timeDelta=3), dict(confidence=2.0, timeDelta=3), None),
}
WEIGHTS = {"depth_cut_2m": 0.5}   # processFrame's weightMultiplier (ElasticFusion.h:70-75)


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_tracker_configurations_match_oracle(hip, seq, name):
    """Every tracker knob of the reference's front-end (MainController flags -fo / -nso / -ftf / -i, setPyramid, setRgbOnly:
    Core/ElasticFusion.h:135-183): 6 free-running frames, tracker statistics, float pose and the whole map identical to the
    oracle run with the same knob (the rgbOnly variant exercises the per-level "break" bookkeeping)."""
    hip_kw, oracle_kw, setter = VARIANTS[name]
    ef = hip.ElasticFusion(**hip_kw)
    if setter:
        setter(ef)
    o = efo.Fusion(**oracle_kw)
    for k in range(6):
        rgb, depth, _ = seq.frame(k)
        wm = WEIGHTS.get(name, 1.0)
        ef.processFrame(rgb, depth, k * 33333, weightMultiplier=wm)
        o.process_frame(rgb, depth, k * 33333, weight=wm)
        if k > 0:
            st, _, _ = ef.trackingStats()
            a, b = np.asarray(st, np.float32), np.asarray(o.stats(), np.float32)
            nan = np.isnan(b)   # lastICPError with the ICP term off: sqrt(0)/0 (the NaN's sign bit differs between x86 and the GPU)
            assert np.array_equal(np.isnan(a), nan) and np.array_equal(a[~nan].view(np.uint32), b[~nan].view(np.uint32)), (name, k, st, o.stats())
        assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32)), (name, k)
        assert ef.lastCount() == o.map_count(), (name, k)
    assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32)), name
    ef.close()


def test_noisy_depth_sequence_matches_oracle(hip):
    """sensor-like depth noise (sigma 1.2 mm x z^2): holes, gate rejections and unstable surfels get exercised"""
    from elasticfusion_amd import synth
    s = synth.Sequence(0xEF0005, noise=True)
    ef, o = hip.ElasticFusion(), efo.Fusion()
    for k in range(8):
        rgb, depth, _ = s.frame(k)
        ef.processFrame(rgb, depth, k * 33333)
        o.process_frame(rgb, depth, k * 33333)
        assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32)), k
        assert ef.lastCount() == o.map_count(), k
    assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32))
    ef.close()


def test_device_resident_frames_match_oracle(hip, seq):
    """ef_process_frame_dev (frames already in HBM, the bench path): the copies are folded into the first kernels that read
    the frame.  Same trajectory and map as the oracle, bit for bit, and the caller's buffers may be reused right after a
    synchronize."""
    n = 8
    o = efo.Fusion()
    ef = hip.ElasticFusion()
    bufs = [(hip.DevBuf.from_array(np.ascontiguousarray(seq.frame(k)[0])), hip.DevBuf.from_array(np.ascontiguousarray(seq.frame(k)[1])))
            for k in range(n)]
    for k in range(n):
        rgb, depth, _ = seq.frame(k)
        o.process_frame(rgb, depth, k * 33333)
        ef.processFrameDevice(bufs[k][0].p.value, bufs[k][1].p.value, k * 33333)
    assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32))
    assert ef.lastCount() == o.map_count()
    assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32))
    assert np.array_equal(ef.image("fill_image"), o.buffer("fill_image"))
    ef.close()


def test_graph_replayed_tracker_matches_oracle(hip, seq):
    """BASELINE.json configs[4]: the tracker's ~70 launches captured into a hipGraph (one per pyramid parity) and replayed.
    Trajectory, tracker statistics and map must be the oracle's bit for bit, as without the graph."""
    n = 10
    o = efo.Fusion()
    ef = hip.ElasticFusion()
    ef.setGraphReplay(True)
    for k in range(n):
        rgb, depth, _ = seq.frame(k)
        o.process_frame(rgb, depth, k * 33333)
        ef.processFrame(rgb, depth, k * 33333)
        if k in (1, 2, 5, n - 1):   # capture frames (1, 2) and replays
            st, _, _ = ef.trackingStats()
            assert np.array_equal(np.asarray(st, np.float32).view(np.uint32), np.asarray(o.stats(), np.float32).view(np.uint32)), k
            assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32)), k
    assert ef.lastCount() == o.map_count()
    assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32))
    ef.close()


def test_frames_at_1280x960_match_oracle(hip):
    """BASELINE.json configs[2] shape: 1280x960, the first frame seeds ~1.2 M surfels.  Same bar as at 640x480: tracker
    statistics, float pose and every surfel identical to the oracle's (the kernels take the resolution at run time)."""
    from elasticfusion_amd import synth
    w, h = 1280, 960
    s = synth.Sequence(0xEF0003, width=w, height=h)
    ef = hip.ElasticFusion(width=w, height=h, fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy, maxSurfels=8 * 1024 * 1024)
    o = efo.Fusion(width=w, height=h, fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy)
    for k in range(3):
        rgb, depth, _ = s.frame(k)
        ef.processFrame(rgb, depth, k * 33333)
        o.process_frame(rgb, depth, k * 33333)
        if k > 0:
            st, _, _ = ef.trackingStats()
            assert np.array_equal(np.asarray(st, np.float32).view(np.uint32), np.asarray(o.stats(), np.float32).view(np.uint32)), k
        assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32)), k
    assert ef.lastCount() == o.map_count() > 1000000
    assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32))
    ef.close()


@pytest.mark.parametrize("size", [(320, 240), (332, 252), (100, 76)])
def test_small_and_odd_resolutions_match_oracle(hip, size):
    """Sizes whose pyramid levels are smaller than one pass of the 16384 virtual threads (80x60, 25x19), whose pixel counts are not
    multiples of it, and whose levels have odd dimensions (83x63): partial passes, idle virtual warps, ragged tiles."""
    from elasticfusion_amd import synth
    W, H = size
    sq = synth.Sequence(seed=0xEF0003, width=W, height=H)
    kw = dict(width=W, height=H, fx=sq.fx, fy=sq.fy, cx=sq.cx, cy=sq.cy)
    ef = hip.ElasticFusion(maxSurfels=1 << 19, **kw)
    o = efo.Fusion(maxSurfels=1 << 19, **kw)
    for k in range(6):
        rgb, depth, _ = sq.frame(k)
        ef.processFrame(rgb, depth, k * 33333)
        o.process_frame(rgb, depth, k * 33333)
        if k > 0:
            st, _, _ = ef.trackingStats()
            assert np.array_equal(np.asarray(st, np.float32).view(np.uint32), np.asarray(o.stats(), np.float32).view(np.uint32)), (k, st, o.stats())
        assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32)), k
        assert ef.lastCount() == o.map_count(), k
    assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32))
    for name, oname in (("image", "image"), ("time", "time"), ("fill_image", "fill_image"), ("index", "index")):
        assert np.array_equal(ef.image(name), o.buffer(oname)), name
    ef.close()


def test_degenerate_frames_match_oracle(hip):
    """An all-invalid depth image, a black colour image and a depth image entirely beyond the cut-off in the middle of a sequence:
    empty vertex maps, zero correspondences (0 / 0 statistics), singular normal equations — the tracker must carry on exactly as
    the oracle does, NaNs included."""
    from elasticfusion_amd import synth
    W, H = 320, 240
    sq = synth.Sequence(seed=0xEF0003, width=W, height=H)
    kw = dict(width=W, height=H, fx=sq.fx, fy=sq.fy, cx=sq.cx, cy=sq.cy)
    ef = hip.ElasticFusion(maxSurfels=1 << 19, **kw)
    o = efo.Fusion(maxSurfels=1 << 19, **kw)
    nans = 0
    for k in range(9):
        rgb, depth, _ = sq.frame(k)
        if k == 3:
            depth = np.zeros_like(depth)
        if k == 5:
            rgb = np.zeros_like(rgb)
        if k == 6:
            depth = np.full_like(depth, 65535)
        ef.processFrame(rgb, depth, k * 33333)
        o.process_frame(rgb, depth, k * 33333)
        if k > 0:
            st, _, _ = ef.trackingStats()
            a, b = np.asarray(st, np.float32), np.asarray(o.stats(), np.float32)
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32)), (k, a, b)
            nans += int(np.isnan(b).any())
        assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32)), k
        assert ef.lastCount() == o.map_count(), k
    assert nans >= 2 and np.isfinite(o.pose()).all()
    assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32))
    ef.close()


def test_fusion_with_injected_poses(hip, seq):
    """Ground-truth poses injected (in_T_wc), the reference's own way of decoupling fusion from tracking
    (ElasticFusion.cpp:302,367-369): identical poses => the map must match surfel for surfel."""
    n = 10
    ef = hip.ElasticFusion(confidence=1.0)
    o = efo.Fusion(confidence=1.0)
    for k in range(n):
        rgb, depth, Tgt = seq.frame(k)
        T = None if k == 0 else Tgt
        ef.processFrame(rgb, depth, k, in_T_wc=T)
        o.process_frame(rgb, depth, k, T_wc=T)
        assert ef.lastCount() == o.map_count(), k
    m, mr = ef.downloadMap(), o.map()
    assert m.shape == mr.shape
    frac = compare_maps(m, mr, rel=1e-5)
    assert frac == 1.0, frac
    exact = float((m.view(np.uint32) == mr.view(np.uint32)).all(axis=1).mean())
    assert exact > 0.999, exact
    for name in ("image", "time", "fill_image"):
        assert np.array_equal(ef.image(name), o.buffer(name)), name
    ef.close()


def test_map_upload_tick_trajectory_timings_and_dumps(hip, seq, tmp_path):
    """The state API around processFrame: a map downloaded from one context and uploaded into a fresh one (ef_map_upload +
    ef_set_tick) continues exactly like the original; the device-resident trajectory log; the per-stage timers; the dumps."""
    # The velocity weighting of a frame (ElasticFusion.cpp:369-383) depends on the PREVIOUS pose, which a restored context does
    # not have; from the restore point on the camera therefore moves fast enough (3 sequence frames per step) for the weight to
    # sit at its floor in both contexts.
    frames = [seq.frame(k) for k in (0, 1, 2, 3, 4, 5, 8, 11, 14)]

    def feed(ef, k):
        rgb, depth, T = frames[k]
        ef.processFrame(rgb, depth, k * 33333, in_T_wc=None if k == 0 else T)

    a = hip.ElasticFusion(confidence=1.0)
    for k in range(6):
        feed(a, k)
    m, tick = a.downloadMap(), a.getTick()
    assert tick == 7 and len(m) == a.lastCount()
    b = hip.ElasticFusion(confidence=1.0)
    b.uploadMap(m)
    b.setTick(tick)
    assert b.getTick() == tick and np.array_equal(b.downloadMap().view(np.uint32), m.view(np.uint32))
    for k in (6, 7):
        feed(a, k)
        feed(b, k)
    assert np.array_equal(a.downloadMap().view(np.uint32), b.downloadMap().view(np.uint32))
    assert np.array_equal(a.image("image"), b.image("image")) and np.array_equal(a.image("time"), b.image("time"))
    b.predict()                                                   # ElasticFusion::predict(): idempotent on an unchanged map and pose
    assert np.array_equal(a.image("image"), b.image("image"))
    b.close()
    Ts, ts = a.trajectory()
    assert len(Ts) == 8 and np.array_equal(ts, np.arange(8) * 33333)
    assert np.array_equal(Ts[-1], a.get_T_wc()) and np.array_equal(Ts[0], np.eye(4))
    for k in range(1, 8):
        assert np.allclose(Ts[k], frames[k][2], atol=1e-12), k
    a.enableTiming(True)
    feed(a, 8)
    a.synchronize()
    t = a.timings()
    for stage in ("Preprocess", "indexMap", "Fuse::Data+Update", "Fuse::Copy", "IndexMap::ACTIVE"):
        assert stage in t and 0 < t[stage] < 50, (stage, t)
    a.enableTiming(False)
    a.saveFreiburg(str(tmp_path / "t.freiburg"))
    a.savePly(str(tmp_path / "m.ply"))
    assert len(open(tmp_path / "t.freiburg").read().splitlines()) == 9
    hdr = open(tmp_path / "m.ply", "rb").read(300)
    n_stable = int((a.downloadMap()[:, 3] > 1.0).sum())
    assert hdr.startswith(b"ply") and (b"element vertex %d" % n_stable) in hdr and n_stable > 10000
    a.close()


def test_two_contexts_interleaved(hip):
    """Nothing in the engine is global (the reference's Resolution / Intrinsics singletons became ef_config fields): two contexts
    of different resolutions in one process, fed alternately, each on its own stream, give what each gives alone."""
    from elasticfusion_amd import synth
    cfgs = [dict(width=320, height=240), dict(width=332, height=252)]
    seqs = [synth.Sequence(seed=0xEF0004 + i, **c) for i, c in enumerate(cfgs)]
    kws = [dict(fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy, maxSurfels=1 << 19, **c) for s, c in zip(seqs, cfgs)]
    n = 6
    solo = []
    for s, kw in zip(seqs, kws):
        ef = hip.ElasticFusion(**kw)
        for k in range(n):
            rgb, depth, _ = s.frame(k)
            ef.processFrame(rgb, depth, k)
        solo.append((ef.get_T_wc(), ef.downloadMap()))
        ef.close()
    efs = [hip.ElasticFusion(**kw) for kw in kws]
    assert efs[0].stream() != efs[1].stream()
    for k in range(n):
        for s, ef in zip(seqs, efs):
            rgb, depth, _ = s.frame(k)
            ef.processFrame(rgb, depth, k)
    for ef, (T, m) in zip(efs, solo):
        assert np.array_equal(ef.get_T_wc(), T)
        assert np.array_equal(ef.downloadMap().view(np.uint32), m.view(np.uint32))
        ef.close()


def test_capacity_overflow_is_reported(hip):
    """max_surfels = width * height holds the first frame exactly; the surfels later frames add do not fit: the map is clamped,
    nothing is written out of bounds, and ef_synchronize says so ONCE (EF_ECAPACITY is a warning: it is cleared when read, and the
    clamped map stays readable — count, download, PLY dump — as the reference's overflowing buffer would)."""
    from elasticfusion_amd import synth
    W, H = 320, 240
    sq = synth.Sequence(seed=0xEF0003, width=W, height=H)
    ef = hip.ElasticFusion(width=W, height=H, fx=sq.fx, fy=sq.fy, cx=sq.cx, cy=sq.cy, maxSurfels=W * H, confidence=1.0)
    rgb, depth, _ = sq.frame(0)
    depth = np.where(depth == 0, 1500, depth).astype(np.uint16)        # every pixel seeds a surfel: the map is full
    ef.processFrame(rgb, depth, 0)
    assert ef.lastCount() == W * H
    for k in range(1, 4):
        rgb, depth, _ = sq.frame(20 * k)
        ef.processFrame(rgb, depth, k)
    with pytest.raises(hip.EFError, match="capacity"):
        ef.synchronize()
    ef.synchronize()                                                   # reported once, then cleared
    assert ef.lastCount() == W * H                                     # clamped, readable
    m = ef.downloadMap()
    assert m.shape == (W * H, 12) and np.isfinite(m[:, :3]).all()
    rgb, depth, _ = sq.frame(80)
    ef.processFrame(rgb, depth, 4)                                     # overflows again: reported again
    with pytest.raises(hip.EFError, match="capacity"):
        ef.synchronize()
    ef.close()


def test_trajectory_log_grows_without_bound(hip):
    """t_T_wc has no capacity in the reference; the device-resident log starts at 1024 poses and doubles on demand."""
    from elasticfusion_amd import synth
    W, H = 100, 76
    sq = synth.Sequence(seed=0xEF0003, width=W, height=H)
    ef = hip.ElasticFusion(width=W, height=H, fx=sq.fx, fy=sq.fy, cx=sq.cx, cy=sq.cy, maxSurfels=1 << 16)
    rgb, depth, _ = sq.frame(0)
    n = 2100
    for k in range(n):
        T = np.eye(4)
        T[0, 3] = 1e-4 * k
        ef.processFrame(rgb, depth, k * 1000, in_T_wc=None if k == 0 else T)
    Ts, ts = ef.trajectory()
    assert len(Ts) == n and np.array_equal(ts, np.arange(n) * 1000)
    assert np.allclose(Ts[:, 0, 3], 1e-4 * np.arange(n), atol=1e-12) and np.array_equal(Ts[1500][:3, :3], np.eye(3))
    ef.close()


def test_context_used_from_another_thread(hip, seq, tmp_path):
    """Every entry point binds the context's device itself (ef_config.device), so a context may be driven by a thread that never
    called hipSetDevice; the results are those of the creating thread.  Run in a child process (with a time limit) so that whatever
    the HIP runtime does with a finished thread's state at exit cannot hold up the test session."""
    import subprocess
    import sys
    code = """
import sys, threading
import numpy as np
sys.path.insert(0, %r)
from elasticfusion_amd import api, synth
seq = synth.Sequence(0xEF0001)
frames = [seq.frame(k) for k in range(3)]
ef = api.ElasticFusion()
out = {}
def work():
    try:
        for k, (rgb, depth, _) in enumerate(frames):
            ef.processFrame(rgb, depth, k)
        out["T"] = ef.get_T_wc(); out["n"] = ef.lastCount()
    except Exception as e:
        out["err"] = repr(e)
t = threading.Thread(target=work); t.start(); t.join()
assert "err" not in out, out
ref = api.ElasticFusion()
for k, (rgb, depth, _) in enumerate(frames):
    ref.processFrame(rgb, depth, k)
assert np.array_equal(out["T"], ref.get_T_wc()) and out["n"] == ref.lastCount()
ef.close(); ref.close()
print("THREAD_OK", out["n"], flush=True)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert "THREAD_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_api_errors(hip):
    with pytest.raises(hip.EFError):
        hip.ElasticFusion(width=642)
    with pytest.raises(hip.EFError):
        hip.ElasticFusion(maxSurfels=1000)


def test_reference_download_mode_matches_the_reference_buffer_choice(hip, seq, tmp_path):
    """ef_set_reference_download(1): ef_map_download / ef_save_ply return what GlobalModel::downloadMap reads in the reference —
    the buffer the frame's update pass wrote (the map before clean) truncated to the count after clean (quirk Q14; observed from
    the compiled reference in tests/test_oracle_vs_reference_frame.py) — frame after frame the oracle's copy of that buffer."""
    import ctypes as C
    ef = hip.ElasticFusion(confidence=1.0)
    ef.setReferenceDownload(True)
    o = efo.Fusion(confidence=1.0)
    differs = 0
    for k in range(6):
        rgb, depth, T = seq.frame(k)
        ef.processFrame(rgb, depth, k * 33333, in_T_wc=None if k == 0 else T)
        o.process_frame(rgb, depth, k * 33333, T_wc=None if k == 0 else T)
        got, want = ef.downloadMap(), o.map_reference()
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), k
        differs += int(not np.array_equal(want, o.map()))
    assert differs >= 4                                   # from the second frame on the two buffers are not the same thing
    ef.savePly(str(tmp_path / "run.ply"))
    lib = hip.lib()
    want = o.map_reference()
    assert lib.ef_write_ply(str(tmp_path / "want.ply").encode(), want.ctypes.data_as(C.c_void_p), C.c_uint(len(want)), C.c_float(1.0)) == 0
    assert open(tmp_path / "run.ply", "rb").read() == open(tmp_path / "want.ply", "rb").read()
    ef.setReferenceDownload(False)                        # back to model(): the map as it stands
    assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32))
    ef.close()
