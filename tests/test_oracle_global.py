"""The global loop closure in the oracle's frame loop (oracle/efo_frame.cpp fernClosure, ElasticFusion.cpp:392-445,609-618) on rendered
frames, with the product's host-side pieces plugged in where Deformation::constrain stands (ef_solve_deformation on every 5th node of
the sampled graph): a view is stored as a keyframe, revisited 400 ticks later with 19 cm of drift — the fern database proposes the
keyframe, the 1/8-resolution ICP registers the view to it, the constraints with their pins reach the optimiser; a map of a few frames
cannot absorb that much drift within the reference's acceptance thresholds, so the closure is rejected and the frame goes on to the
local closure with its pose unchanged.  (The accepted flow: tests/test_oracle_vs_reference_frame.py::test_accepted_global_closure_flow;
the optimiser accepting: tests/test_deform_global_vs_reference.py.)"""
import ctypes as C

import numpy as np

import efo
from elasticfusion_amd import api, synth


def test_rendered_revisit_is_matched_registered_and_gated():
    seq = synth.Sequence(seed=0xEF0001)
    o = efo.Fusion(timeDelta=200, confidence=2.0)
    o.set_close_loops(True)
    o.enable_ferns(seed=7)
    calls = []

    def solver(fernMatch, rows, poses, times):
        nodes = efo.sample_graph(o.map())
        if fernMatch:
            nodes = nodes[::5]                                   # Deformation::sampleGraphFrom
        rr = [(r[0:3], r[3:6], int(r[6]), int(r[7]), int(r[8]), int(r[9])) for r in rows]
        got = api.solve_deformation(nodes, rr, fernMatch, 0, poses, times)
        calls.append((fernMatch, len(rows), len(nodes), got["accepted"], got["error"], got["meanConsErr"]))
        if not got["accepted"]:
            return None
        rel = np.array([list(a) + list(b) + [c, d, 1, 0] for a, b, c, d, _, _ in got["new_relative"]]).reshape(-1, 10)
        return dict(graph=got["graph"], poses=got["poses"], new_relative=rel)

    o.set_deform_solver(solver)
    kept = []
    for k in range(6):
        rgb, depth, T = seq.frame(k)
        o.process_frame(rgb, depth, k, T_wc=T)
        kept.append(len(o.ferns()))
        assert o.global_loop().closest == -1                     # nothing is older than 300 ticks yet (Ferns.cpp:218)
    assert kept[0] == 1 and kept[-1] >= 1 and kept == sorted(kept)   # the first frame is always a keyframe; near-duplicates are not
    assert not calls                                             # no closure attempted: neither a fern match nor open local gates
    o.set_tick(o.tick() + 400)
    d = np.eye(4)
    a = 0.02
    d[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
    d[:3, 3] = [0.16, -0.05, 0.09]
    rgb, depth, T = seq.frame(3)
    efo.lib().efo_fusion_trace(o.h_, 1)
    o.process_frame(rgb, depth, 100, T_wc=T @ d)
    take = efo.lib().efo_fusion_take_trace
    take.restype = C.c_char_p
    trace = take(o.h_).decode().splitlines()
    g = o.global_loop()
    assert g.attempted and 0 <= g.closest < kept[-1]
    assert g.icp_error < 3e-4 and g.icp_count > 2400 and 30 <= g.n_constraints <= 50         # the gates of Ferns.cpp:263-264
    rec = np.array(g.T_wc_recovery).reshape(4, 4)
    stored = o.ferns().frame(g.closest)["T_wc"]
    assert np.abs(rec[:3, 3] - T[:3, 3]).max() < 0.01 < np.abs((T @ d)[:3, 3] - T[:3, 3]).max()   # registered to the old view: the drift is gone
    assert np.abs(rec[:3, 3] - stored[:3, 3]).max() < 0.05                                     # ... starting from the keyframe's pose
    fm, n_rows, n_nodes, accepted, error, mean = calls[0]
    assert fm and n_rows == 2 * g.n_constraints and n_nodes > 4
    assert mean > 0.06 and not accepted and error > 0.12 and not g.accepted                    # entered (Deformation.cpp / DeformationGraph.cpp:425), rejected (:153)
    assert np.abs(o.pose() - T @ d).max() < 1e-12                                              # the pose stays
    i = trace.index("global.constrain fernMatch=1 constraints=%d relative=0" % n_rows)
    assert trace[i + 1].startswith("combinedPredict INACTIVE")                                 # ... and the local closure gets its turn (:447)
    assert any(l.startswith("clean ") and l.endswith("nodes=0 timeDelta=200 maxDepth=20 isFern=0") for l in trace)
