"""Relocalisation (the reference constructor's `reloc`; ElasticFusion.cpp:326-366, 402-413, 536, 601-604, 624-649) in the oracle's frame
loop, pinned against the COMPILED reference frame loop (oracle/_ref/libefr_frame.so: Core/ElasticFusion.cpp itself over the GL tape
recorder and a scripted tracker): the oracle runs a real sequence — good frames, then frames whose depth is a small patch (the tracker's
covariance blows up), then a view it has a keyframe of — and after every frame the reference's tracker double is scripted with the
oracle tracker's verdicts (ICP error, covariance) of that frame; everything that FOLLOWS from the verdicts must then agree step by step:
which frames are fused, the counter towards "lost", the raw-frame fill-in and the frozen tick while lost, the fern match taken as the
pose, the whole-model prediction (time = 0) for the probation frame, and the way back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import efo
import fernscene
from test_oracle_vs_reference_frame import H, W, P, Ref, have, lib, oracle_lines, translate

pytestmark = pytest.mark.skipif(not have(), reason="oracle/_ref/libefr_frame.so can only be built where /root/reference exists")

TD, CONF = 200, 2.0
N_GOOD, N_BAD = 4, 14


def patch_only(depth):
    """a frame whose depth is one small patch: a handful of correspondences on a near-planar piece, no constraint on most of the pose"""
    bad = np.zeros_like(depth)
    bad[200:260, 280:360] = depth[200:260, 280:360]
    return bad


def scenario(seq):
    """(rgb, depth, what) per frame: N_GOOD good frames (keyframes are stored), the tick advanced by 400 (a keyframe must be older than
    300 ticks to be proposed), N_BAD patch-only frames (lost after the eleventh that is not ok), then the view of frame 3 three times"""
    out = []
    for k in range(N_GOOD):
        r, d, _ = seq.frame(k)
        out.append((r, d, "good"))
    r, d, _ = seq.frame(N_GOOD - 1)
    for _ in range(N_BAD):
        out.append((r, patch_only(d), "bad"))
    r, d, _ = seq.frame(N_GOOD - 1)
    for _ in range(3):
        out.append((r, d, "back"))
    return out


def wild(lines):
    """what cannot agree between a scripted and a real tracker: poses, the model-or-fill-in choice (denseEnough reads pixels), the
    velocity weighting, constraint counts; and the oracle-only lines"""
    out = []
    for l in lines:
        # (the optimiser calls are pinned by test_accepted_*_flow; here the reference's graphs are never initialised: no node read-backs scripted)
        if l.startswith(("  pose", "ferns.", "local.constrain", "global.constrain", "denseEnough")):
            continue
        l = re.sub(r"(vertices|normals|image)=(fill|pred)", r"\1=*", l)
        l = re.sub(r"(constraints|weighting|relative)=\S+", r"\1=*", l)
        out.append(l)
    return out


@pytest.mark.parametrize("mode", ["close_loops", "open_loop"])
def test_relocalisation_flow_matches_the_compiled_reference(tmp_path, seq, mode):
    """close_loops: lost and found again through the fern database.  open_loop (the reference's -o -rl): the same verdicts, but nothing
    ever looks for the way back — the camera stays lost, the tick frozen, nothing fused, whatever the frames show afterwards."""
    close = mode == "close_loops"
    so = lib()
    so.efe_queue_readpixels.argtypes = [P, C.c_long]
    so.efe_set_tick.argtypes = [P, C.c_int]
    so.efe_lost.argtypes = [P]
    so.efe_ferns_last_closest.argtypes = [P]
    efo.lib().efo_set_threads(min(16, os.cpu_count() or 1))
    o = efo.Fusion(timeDelta=TD, confidence=CONF)
    if close:
        o.set_close_loops(True)
        o.enable_ferns(seed=7)
        o.set_deform_solver(lambda *a: None)                   # the optimiser rejects (as the reference's scripted one does below)
    o.set_reloc(True)
    efo.lib().efo_fusion_trace(o.h_, 1)
    take = efo.lib().efo_fusion_take_trace
    take.restype = C.c_char_p
    ref = Ref(so, str(tmp_path / "ref"), timeDelta=TD, closeLoops=3 if close else 2, confidence=CONF)     # bit 0: closeLoops, bit 1: reloc
    key = fernscene.place(2)
    blank = tuple(np.zeros_like(a) for a in key)
    dense = np.zeros(32 * 24 * 3, np.uint8)                    # Resize::image for denseEnough (ElasticFusion.cpp:304): 1/20 resolution

    def queue(*items):
        so.efe_clear_queues()
        for a in items:
            so.efe_queue_readpixels(a.ctypes.data, a.nbytes)

    history = []
    frames = scenario(seq)
    for k, (rgb, depth, what) in enumerate(frames):
        if k == N_GOOD:
            o.set_tick(o.tick() + 400)
            so.efe_set_tick(ref.h, so.efe_tick(ref.h) + 400)
        o.process_frame(rgb, depth, k * 33333)
        st = o.reloc_state()
        g = o.global_loop()
        history.append((what, st["lost"], st["trackingOk"], st["trackingCount"], st["lastFrameRecovery"], g.closest, o.tick()))
        got = wild(oracle_lines(take(o.h_).decode()))
        # the oracle tracker's verdicts of this frame -> the reference's tracker double
        if k > 0:
            stats, A, _ = o.odometry().stats()
            cov = efo.covariance(A)
            cov_bad = bool((np.diag(cov) > 1e-4).any())
            D = np.eye(4)
            D[:3, 3] = [0.001, -0.002, 0.0005]
        matched = bool(g.attempted and g.closest >= 0)
        proposed = any(l.startswith("fernOdom.track") for l in got)     # a keyframe was similar enough to be registered against the view
        if k > 0:
            # one script for the three trackers: 3000 / 1000 correspondences pass / fail the fern gates (2400, 1400 when lost) and keep the
            # local closure's gate (35000) shut
            so.efe_script_tracker(D.ctypes.data, 1e-6 if stats[0] < 1e-4 else 1e-3, 3000.0 if matched else 1000.0, 1e-3 if cov_bad else 1e-7, 0)
        view = key if proposed else blank                      # what Ferns::findFrame reads back: the stored keyframe's view again, or nothing
        if not close:
            queue(dense)                                       # open loop: only denseEnough reads pixels back (the fern database is idle)
        elif k == 0:
            queue(*key)                                        # first frame: Ferns::addFrame only -> keyframe 0
        elif st["lost"]:
            queue(dense, *view)                                # denseEnough, findFrame; a lost camera stores no keyframe (:601-604)
        else:
            queue(dense, *view, *blank)                        # denseEnough, findFrame, addFrame
        want = wild(translate(ref, ref.frame(rgb, depth, k * 33333)))
        assert got == want, (k, what, "\n".join(got), "----", "\n".join(want))
        assert bool(so.efe_lost(ref.h)) == st["lost"], (k, what)
        assert so.efe_tick(ref.h) == o.tick(), (k, what)
        assert not close or (so.efe_ferns_last_closest(ref.h) >= 0) == bool(matched) or k == 0, (k, what)
        if st["lost"]:                                         # (lost at the end of the frame: lost all the way through it)
            assert not any(l.startswith(("fuse", "predictIndices", "clean", "modelToModel")) for l in got), (k, what)
    so.efe_clear_queues()
    ref.close()

    # the story itself, as the oracle lived it
    whats = [h[0] for h in history]
    lost = [h[1] for h in history]
    assert not any(lost[:N_GOOD]) and all(h[2] for h in history[:N_GOOD])
    first_lost = lost.index(True)
    assert whats[first_lost] == "bad" and history[first_lost][3] == 11                    # the eleventh frame in a row that is not ok
    assert all(history[first_lost - j][3] == 11 - j for j in range(1, 11))                # ... counted one by one
    ticks = [h[6] for h in history]
    assert all(ticks[k] == ticks[first_lost - 1] for k in range(first_lost, len(history)) if lost[k])   # the tick stands still while lost
    back = whats.index("back")
    if not close:
        assert all(lost[first_lost:]) and ticks[-1] == ticks[first_lost - 1]                 # lost for good
        return
    assert lost[back] and history[back][5] >= 0 and history[back][4]                     # lost, keyframe matched, pose taken: probation next
    assert not lost[back + 1] and history[back + 1][2] and not history[back + 1][4]      # found again
    assert ticks[back + 1] == ticks[back] + 1 and ticks[back + 2] == ticks[back] + 2
    T = seq.frame(N_GOOD - 1)[2]
    assert np.abs(o.pose()[:3, 3] - T[:3, 3]).max() < 0.02                               # back where the keyframe says the camera is


def test_a_frame_that_is_not_ok_is_not_fused_and_drops_its_deformation(seq):
    """trackingOk = false without being lost (:536): no fusion, and a deformation accepted in that frame never reaches the map — the map
    after the frame is bit for bit the map before it"""
    efo.lib().efo_set_threads(min(16, os.cpu_count() or 1))
    o = efo.Fusion(confidence=CONF)
    o.set_reloc(True)
    for k in range(2):
        r, d, _ = seq.frame(k)
        o.process_frame(r, d, k)
    before = o.map().copy()
    tick = o.tick()
    r, d, _ = seq.frame(2)
    o.set_deformation(np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (6, 1)))
    o.process_frame(r, np.zeros_like(d), 2)                    # the lens covered: no correspondence, lastICPError = 0 / 0, "NaN < 1e-4" is false
    st = o.reloc_state()
    assert np.isnan(o.stats()[0]) and not st["trackingOk"] and st["trackingCount"] == 1 and not st["lost"]
    assert o.tick() == tick + 1                                # not lost: the tick goes on
    assert np.array_equal(o.map().view(np.uint32), before.view(np.uint32))
    r, d, _ = seq.frame(2)
    o.process_frame(r, d, 3)                                   # nothing to track against: a young map's fill-in is the (empty) previous frame
    st = o.reloc_state()
    assert not st["trackingOk"] and st["trackingCount"] == 2 and not st["lost"]
    assert np.array_equal(o.map().view(np.uint32), before.view(np.uint32))
    o.process_frame(r, d, 4)                                   # frame to frame again: ok, fused (the dropped graph does not come back)
    st = o.reloc_state()
    assert st["trackingOk"] and st["trackingCount"] == 0 and o.map_count() != len(before)
    assert o.tick() == tick + 3
