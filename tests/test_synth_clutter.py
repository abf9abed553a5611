"""CPU: the second synthetic scene family (elasticfusion_amd/synth.py::ClutterSequence — planar clutter, thin rods, disparity-quantised depth
with discontinuity / grazing-angle / rectangular drop-outs) is what its docstring says, is reproducible frame by
frame, and is trackable: the oracle follows it through the bootstrap regime about as well as it follows the box-and-spheres scene."""
import numpy as np

import efo


def test_frames_are_reproducible_and_sensor_like():
    from elasticfusion_amd import synth
    a = synth.ClutterSequence(0xEF0001, noise=True)
    b = synth.ClutterSequence(0xEF0001, noise=True)
    rgb5, d5, T5 = a.frame(5)
    _ = b.frame(9)                                   # another rendering order
    rgb5b, d5b, _ = b.frame(5)
    assert np.array_equal(d5, d5b) and np.array_equal(rgb5, rgb5b)
    assert rgb5.min() >= 1                            # 0 means "invalid" to the tracker (reduce.cu:647,679)
    valid = d5 > 0
    assert 0.85 < valid.mean() < 0.99                 # holes: discontinuities, grazing incidence, drop-outs
    assert d5[valid].min() >= 300 and d5[valid].max() <= 3000
    # disparity quantisation: the depth values present are few (1/8 px steps of a 580 px x 75 mm rig), not every millimetre
    assert len(np.unique(d5[valid])) < 0.6 * (int(d5[valid].max()) - int(d5[valid].min()))
    # clutter in front of the room's walls, and thin structures: connected runs of a few pixels that are much nearer than their surroundings
    clean = synth.ClutterSequence(0xEF0001)
    _, d0, _ = clean.frame(0)
    near = (d0 > 0) & (d0 < 1400)
    assert 0.08 < near.mean() < 0.6
    runs = np.diff(np.flatnonzero(np.diff(np.concatenate([[0], near[240].astype(np.int8), [0]])) != 0))[::2]
    assert (runs <= 12).any()                         # a rod crosses the middle row


def test_oracle_tracks_the_clutter_scene():
    from elasticfusion_amd import synth
    efo.set_threads(8)
    s = synth.ClutterSequence(0xEF0003)
    o = efo.Fusion(confidence=2.0)
    worst = 0.0
    for k in range(8):
        rgb, depth, T = s.frame(k)
        o.process_frame(rgb, depth, k * 33333)
        worst = max(worst, float(np.linalg.norm(o.pose()[:3, 3] - T[:3, 3])))
    efo.set_threads(1)
    st = o.stats()
    assert st[1] > 100000 and st[3] > 5000            # ICP inliers and photometric correspondences in the usual range
    assert worst < 0.01, worst                        # within a centimetre of the generating trajectory over the first frames
