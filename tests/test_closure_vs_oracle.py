"""The product's host-side closure object (ef_closure_*, elasticfusion_amd/csrc/ef_ferns.hip: fern database + which constraints reach which
deformation graph + what an accepted closure changes) against the oracle's frame loop (oracle/efo_frame.cpp fernClosure /
localLoopClosure / processFerns, itself equal to the compiled ElasticFusion.cpp).  The oracle processes rendered frames; the closure
object is fed what the device side will feed it — the 1/8-resolution views, the poses, the graph nodes, the fern tracker's answer — and
has to take the same decisions: keyframes kept, fern matched, rows handed to the optimiser, recovered pose, graph, keyframe and
trajectory poses after the deformation, relative constraints kept."""
import os

import numpy as np
import pytest

import efo
import loopscene
from elasticfusion_amd import api, synth


@pytest.fixture(autouse=True)
def oracle_threads():
    """the oracle's frame loop on several host threads (its parallel loops are block-partitioned: same results, a third of the time)"""
    efo.lib().efo_set_threads(min(16, os.cpu_count() or 1))
    yield
    efo.lib().efo_set_threads(1)


def rows_of(rows):
    return [(r[0:3], r[3:6], int(r[6]), int(r[7]), int(r[8]), int(r[9])) for r in rows]


def chain_for(rows, tick, per=5, n=200):
    """a long trajectory's worth of graph nodes for a map that only has a few frames: `n` global nodes (every `per`-th of the local ones)
    from the constraints' targets (old) to their sources (now), so that the reference's acceptance thresholds can be met"""
    plain = rows[(rows[:, 8] == 0) & (rows[:, 9] == 0)]
    a, b = plain[:, 3:6].mean(0), plain[:, 0:3].mean(0)
    s = np.linspace(0, 1, n * per)
    nodes = np.zeros((n * per, 4), np.float32)
    nodes[:, :3] = a + s[:, None] * (b - a) + np.random.RandomState(5).uniform(-0.5, 0.5, (n * per, 3))     # spread over the surface, not a line
    nodes[:, 3] = np.floor(1 + s * (tick - 2))
    return nodes


def test_global_closure_decisions():
    seq = synth.Sequence(seed=0xEF0001)
    o = efo.Fusion(timeDelta=200, confidence=2.0)
    o.set_close_loops(True)
    o.enable_ferns(seed=7)
    c = api.Closure(seed=7)
    assert np.array_equal(c.ferns.conservatory, o.ferns().conservatory)
    seen = {}

    def solver(fernMatch, rows, poses, times):
        assert fernMatch
        nodes = chain_for(rows, o.tick())
        got = api.solve_deformation(nodes[::5], rows_of(rows), True, 0, poses, times)
        seen.update(rows=rows.copy(), nodes=nodes, poses=poses.copy(), times=times.copy(), got=got)
        return dict(graph=got["graph"], poses=got["poses"]) if got["accepted"] else None

    o.set_deform_solver(solver)
    never = lambda *a: (_ for _ in ()).throw(AssertionError("no candidate can be old enough"))
    for k in range(5):
        rgb, depth, T = seq.frame(k)
        o.process_frame(rgb, depth, k, T_wc=T)
        tick = o.tick() - 1
        mid = o.fern_view(0)
        if k > 0:                                                # tick 1 has no mid-frame step (ElasticFusion.cpp:290-296)
            acc, _, g = c.globalClosure(*mid, T, tick, never, np.zeros((0, 4), np.float32))
            assert not acc and len(g) == 0 and c.ferns.lastClosest == -1 == o.global_loop().closest
        c.endFrame(*o.fern_view(1), o.pose(), tick)
        assert len(c.ferns) == len(o.ferns())
    assert not seen
    o.set_tick(o.tick() + 400)
    d = np.eye(4)
    d[:3, 3] = [0.16, -0.05, 0.09]
    rgb, depth, T = seq.frame(2)
    tick = o.tick()
    o.process_frame(rgb, depth, 100, T_wc=T @ d)
    g = o.global_loop()
    assert g.closest >= 0 and g.accepted and seen["got"]["accepted"]         # with a long enough graph the reference's thresholds are met
    rec = np.array(g.T_wc_recovery).reshape(4, 4)
    asked = {}

    def tracker(fv, fn, Tf, cv, cn, Tin):
        asked.update(Tf=Tf.copy(), fv=fv.copy())
        return rec, g.icp_error, g.icp_count                      # what the device's 80x60 ICP will answer: here the oracle's

    acc, Tr, graph = c.globalClosure(*o.fern_view(0), T @ d, tick, tracker, seen["nodes"])
    assert acc and c.ferns.lastClosest == g.closest
    assert np.abs(Tr - rec).max() < 1e-12 and np.abs(o.pose() - rec).max() < 1e-12
    assert np.array_equal(asked["fv"], o.ferns().frame(g.closest)["verts"])
    rows, err, mean = c.lastRows()
    assert rows.shape == seen["rows"].shape and np.abs(rows - seen["rows"]).max() < 1e-12
    assert err < 0.12 and mean < 3e-4                              # Deformation.cpp:154
    assert len(graph) == 200 and np.array_equal(graph, seen["got"]["graph"])
    nf = len(o.ferns())
    for i in range(nf):                                            # keyframe poses and the trajectory were deformed along, identically
        assert np.abs(c.ferns.frame(i)["T_wc"] - o.ferns().frame(i)["T_wc"]).max() < 1e-12
    moved = np.abs(seen["got"]["poses"][:, :3, 3] - seen["poses"][:, :3, 3]).max()
    assert moved > 1e-4
    c.endFrame(*o.fern_view(1), o.pose(), tick)
    assert np.abs(c.trajectory() - o.trajectory()).max() < 1e-12 and len(c.trajectory()) == 6
    assert c.counts() == dict(deforms=0, fernDeforms=1, relative=0, trajectory=6) and len(c.ferns) == len(o.ferns())
    c.close()


def test_local_closure_decisions():
    """the synthetic revisit of tests/loopscene.py: the local gates open, the optimiser runs on the sampled graph, keyframe poses follow,
    a third of the new relative constraints is kept — the closure object next to the oracle's frame loop, closure after closure"""
    o = efo.Fusion(timeDelta=loopscene.TIME_DELTA, confidence=loopscene.CONFIDENCE)
    o.set_close_loops(True)
    o.enable_ferns(seed=11)
    c = api.Closure(seed=11)
    state = dict(last=0)
    log = []

    def solver(fernMatch, rows, poses, times):
        assert not fernMatch
        nodes = efo.sample_graph(o.map())
        got = api.solve_deformation(nodes, rows_of(rows), False, state["last"], poses, times)
        log.append(dict(rows=rows.copy(), nodes=nodes, got=got, tick=o.tick()))
        if not got["accepted"]:
            return None
        state["last"] = o.tick()                                  # Deformation.cpp:199-201
        rel = np.array([list(a) + list(b) + [s, t, 1, 0] for a, b, s, t, _, _ in got["new_relative"]]).reshape(-1, 10)
        return dict(graph=got["graph"], poses=got["poses"], new_relative=rel)

    o.set_deform_solver(solver)
    closures = 0
    for i, (rgb, depth, T) in enumerate(loopscene.frames()):
        before = len(log)
        o.process_frame(rgb, depth, i, T_wc=T)
        tick = o.tick() - 1
        if len(log) > before:                                     # the oracle's gates opened on this frame
            info, cons = o.local_loop()
            e = log[-1]
            acc, graph = c.localClosure(cons, tick, e["nodes"])
            assert acc == e["got"]["accepted"] == bool(info.applied)
            rows, _, _ = c.lastRows()
            assert rows.shape == e["rows"].shape and np.abs(rows - e["rows"]).max() < 1e-12
            if acc:
                closures += 1
                assert np.array_equal(graph, e["got"]["graph"])
        c.endFrame(*o.fern_view(1), o.pose(), tick)
        assert len(c.ferns) == len(o.ferns())
        for k in range(len(c.ferns)):
            assert np.abs(c.ferns.frame(k)["T_wc"] - o.ferns().frame(k)["T_wc"]).max() < 1e-12
    assert closures >= 2
    rel = c.relativeConstraints()
    assert len(rel) >= 3 and np.abs(rel - o.relative_constraints()).max() < 1e-12 and set(rel[:, 8]) == {1.0}
    assert c.counts()["deforms"] == closures and c.counts()["fernDeforms"] == 0
    c.close()


def test_lost_camera_decisions():
    """relocalisation (ElasticFusion.cpp:395-413, 601-604 with lost = true): while the oracle's camera is lost the closure object is fed
    what the device side feeds it — the mid-frame view (the raw frame: pass-through fill-in), the pose, the fern tracker's answer — and
    has to decide like the oracle: no match on the patch-only frames, the keyframe matched (count gate 1400) when the known view is back,
    the registered pose handed over; the pose joins the trajectory every frame, no keyframe is stored while lost"""
    from test_oracle_reloc import CONF, N_GOOD, TD, scenario
    seq = synth.Sequence(seed=0xEF0001)
    o = efo.Fusion(timeDelta=TD, confidence=CONF)
    o.set_close_loops(True)
    o.enable_ferns(seed=7)
    o.set_reloc(True)
    o.set_deform_solver(lambda *a: None)
    c = api.Closure(seed=7)
    frames = scenario(seq)
    found_while_lost = 0
    for k, (rgb, depth, what) in enumerate(frames):
        if k == N_GOOD:
            o.set_tick(o.tick() + 400)
        tick = o.tick()
        o.process_frame(rgb, depth, k * 33333)
        st, g = o.reloc_state(), o.global_loop()
        lost_mid = st["lost"]                  # `lost` is set (:343) or cleared (:359) before the mid-frame step, never after it
        rec = np.array(g.T_wc_recovery).reshape(4, 4)
        answer = lambda fv, fn, Tf, cv, cn, Tin: (rec, g.icp_error, g.icp_count)      # the device's 80x60 registration: here the oracle's
        if k > 0 and g.attempted:
            # the pose findFrame saw: the tracker's estimate, which the recovery (if any) then replaced
            T_mid = o.trajectory()[-1] if not (lost_mid and g.closest >= 0) else None
            if lost_mid:
                ok, Tr = c.relocalise(*o.fern_view(0), T_mid if T_mid is not None else rec, tick, answer)
                assert ok == (g.closest >= 0), (k, what)
                assert c.ferns.lastClosest == g.closest, (k, what)
                if ok:
                    found_while_lost += 1
                    assert np.abs(Tr - rec).max() < 1e-12 and np.abs(o.pose() - rec).max() < 1e-12 and st["lastFrameRecovery"]
            else:
                acc, _, _ = c.globalClosure(*o.fern_view(0), T_mid, tick, answer, np.zeros((0, 4), np.float32))
                assert not acc and c.ferns.lastClosest == g.closest, (k, what)
        if st["lost"]:
            c.logPose(o.pose(), tick)
        else:
            c.endFrame(*o.fern_view(1), o.pose(), tick)
        assert len(c.ferns) == len(o.ferns()), (k, what)
        assert len(c.trajectory()) == len(o.trajectory()) == k + 1
    assert found_while_lost == 1 and not o.reloc_state()["lost"]
    assert np.abs(c.trajectory() - o.trajectory()).max() < 1e-12
    c.close()
