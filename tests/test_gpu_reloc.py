"""GPU: relocalisation (ef_set_relocalisation = the reference constructor's `reloc`; ElasticFusion.cpp:326-366, 402-413, 536, 601-604,
624-649) against the oracle's frame loop, which tests/test_oracle_reloc.py pins step by step against the compiled ElasticFusion.cpp:
the tracker's verdict on itself read back every frame, frames that are not ok left unfused, the camera lost after the eleventh in a row,
the raw-frame fill-in and the frozen tick while lost, the fern match (1/8-resolution registration on the device, count gate 1400) taken
as the pose, the whole-model prediction for the probation frame, found again — same states, same poses, same map, frame by frame."""
import numpy as np
import pytest

import efo
from test_gpu_global import oracle_with_ferns, same_frame_state
from test_oracle_reloc import CONF, N_GOOD, TD, scenario

pytestmark = pytest.mark.gpu


def same_reloc_state(ef, o, k):
    s, so = ef.relocState(), o.reloc_state()
    assert (bool(s.lost), bool(s.tracking_ok), s.tracking_count, bool(s.last_frame_recovery)) == \
        (so["lost"], so["trackingOk"], so["trackingCount"], so["lastFrameRecovery"]), k
    assert ef.getTick() == o.tick(), k
    assert ef.getLost() == so["lost"], k


def test_lost_and_found_matches_oracle(seq):
    from elasticfusion_amd import api
    seed = 7
    ef = api.ElasticFusion(closeLoops=True, timeDelta=TD, confidence=CONF, reloc=True)
    ef.useBuiltinLoopSolver(True)
    ef.enableGlobalClosure(seed=seed)
    o, _ = oracle_with_ferns(seed, timeDelta=TD, confidence=CONF)
    o.set_reloc(True)
    frames = scenario(seq)
    lost, closest, ticks = [], [], []
    for k, (rgb, depth, what) in enumerate(frames):
        if k == N_GOOD:
            ef.setTick(ef.getTick() + 400)
            o.set_tick(o.tick() + 400)
        ef.processFrame(rgb, depth, k * 33333)
        o.process_frame(rgb, depth, k * 33333)
        same_reloc_state(ef, o, k)
        same_frame_state(ef, o, k)
        if k > 0:                                        # the verdict's inputs: the same statistics (NaN where there was no correspondence)
            assert np.array_equal(ef.trackingStats()[0], o.stats(), equal_nan=True), k
        g, go = ef.globalLoop(), o.global_loop()
        if g.closest >= 0:                               # the 1/8-resolution registration on the device: same statistics, same pose
            assert np.float32(g.icp_error).view(np.uint32) == np.float32(go.icp_error).view(np.uint32) and g.icp_count == go.icp_count, k
            assert np.array_equal(np.array(g.T_wc_recovery, np.float64).astype(np.float32), np.array(go.T_wc_recovery, np.float64).astype(np.float32)), k
        lost.append(ef.getLost())
        closest.append(g.closest)
        ticks.append(ef.getTick())
    # the story: lost during the patch-only frames, found through a keyframe when the known view comes back
    first_lost = lost.index(True)
    back = N_GOOD + sum(1 for f in frames if f[2] == "bad")
    assert N_GOOD + 10 <= first_lost < back and all(lost[first_lost:back + 1]) and not any(lost[back + 1:])
    assert closest[back] >= 0 and len(set(ticks[first_lost:back + 1])) == 1 and ticks[-1] == ticks[back] + 2
    T = seq.frame(N_GOOD - 1)[2]
    assert np.abs(ef.get_T_wc()[:3, 3] - T[:3, 3]).max() < 0.02
    # one logged pose per processed frame, lost or not; the closure object's trajectory too
    poses, stamps = ef.trajectory()
    assert len(poses) == len(frames) and list(stamps) == [k * 33333 for k in range(len(frames))]
    assert len(ef.closure().trajectory()) == len(frames) == len(o.trajectory())
    assert np.array_equal(ef.closure().trajectory().astype(np.float32), np.array(o.trajectory()).astype(np.float32))
    assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32))
    ef.close()


def test_covered_lens_frames_are_not_fused(seq):
    """no correspondence at all: lastICPError = 0 / 0 and "NaN < 1e-4" is false — the frame is not ok, is not fused, and the deformation
    handed over for it is dropped; the tick goes on (not lost)"""
    from elasticfusion_amd import api
    ef = api.ElasticFusion(confidence=CONF, reloc=True)
    o = efo.Fusion(confidence=CONF)
    o.set_reloc(True)
    for k in range(2):
        r, d, _ = seq.frame(k)
        ef.processFrame(r, d, k)
        o.process_frame(r, d, k)
    before = ef.downloadMap().copy()
    r, d, _ = seq.frame(2)
    graph = np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (6, 1))
    ef.setDeformation(graph)
    o.set_deformation(graph)
    for k, depth in enumerate((np.zeros_like(d), d, d)):
        ef.processFrame(r, depth, 2 + k)
        o.process_frame(r, depth, 2 + k)
        same_reloc_state(ef, o, k)
        assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32)), k
        assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32)), k
        if k < 2:
            assert not ef.relocState().tracking_ok and np.array_equal(ef.downloadMap().view(np.uint32), before.view(np.uint32)), k
    assert ef.relocState().tracking_ok and ef.lastCount() != len(before)
    ef.close()


def test_relocalisation_off_changes_nothing(seq):
    """reloc = false (the default): no read-back, no verdict — bit for bit the frames of a context that never heard of it"""
    from elasticfusion_amd import api
    a = api.ElasticFusion(confidence=CONF)
    b = api.ElasticFusion(confidence=CONF, reloc=True)
    b.setRelocalisation(False)
    for k in range(4):
        r, d, _ = seq.frame(k)
        a.processFrame(r, d if k != 2 else np.zeros_like(d), k)
        b.processFrame(r, d if k != 2 else np.zeros_like(d), k)
        assert not b.getLost() and b.relocState().tracking_ok
    assert np.array_equal(a.downloadMap().view(np.uint32), b.downloadMap().view(np.uint32))
    assert np.array_equal(a.get_T_wc(), b.get_T_wc())
    a.close()
    b.close()
