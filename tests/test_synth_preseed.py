"""synth.sample_surfels (the map bench.py --preseed uploads for BASELINE configs[2], SURVEY 8d: "surfels sampled on the scene surfaces,
radius 4 mm, conf 12") must describe the SAME scene the synthetic frames show: projected through frame 0's camera, the sampled surfels
that face the camera and are not occluded must land on the rendered depth; layout and counts as the engine's map holds them."""
import numpy as np

from elasticfusion_amd import synth


def test_sampled_surfels_lie_on_the_rendered_scene():
    seq = synth.Sequence(0xEF0001, width=320, height=240)
    m = synth.sample_surfels(seq, n=1 << 16)
    assert m.dtype == np.float32 and m.shape[1] == 12 and abs(len(m) - (1 << 16)) < 0.02 * (1 << 16)
    assert (m[:, 3] == 12.0).all() and (m[:, 11] == np.float32(0.004)).all() and (m[:, 5] == 0).all() and (m[:, 6] == 1).all() and (m[:, 7] == 1).all()
    assert np.abs(np.linalg.norm(m[:, 8:11], axis=1) - 1).max() < 1e-5
    rgb_code = m[:, 4].astype(np.int64)
    assert (rgb_code > 0).all() and (rgb_code < (1 << 24)).all()           # color.glsl:19-34 packing, no zero byte triple
    _, depth, T0 = seq.frame(0)
    assert np.abs(T0 - np.eye(4)).max() < 1e-12                            # world frame = camera frame of frame 0
    p = m[:, :3].astype(np.float64)
    z = p[:, 2]
    u = np.rint(p[:, 0] / z * seq.fx + seq.cx).astype(np.int64)
    v = np.rint(p[:, 1] / z * seq.fy + seq.cy).astype(np.int64)
    facing = (m[:, 8:11].astype(np.float64) * p).sum(1) > 0                  # normals point away from the camera (like normals from a depth image)
    ok = (z > 0.3) & (z < 3.0) & (u >= 1) & (v >= 1) & (u < seq.width - 1) & (v < seq.height - 1) & facing
    d = depth[v[ok], u[ok]].astype(np.float64) / 1000.0
    diff = d - z[ok]
    on_surface = np.abs(diff) < 0.02        # the surfel IS the surface the frame shows at that pixel
    occluded = diff <= -0.02                # a sphere stands in front of the wall it lies on
    floating = diff >= 0.02                 # nothing of the scene may hang in free space (a few at a sphere's limb, where the pixel-centre ray passes it)
    assert ok.sum() > 3000 and on_surface.mean() > 0.7 and floating.mean() < 0.02 and (on_surface | occluded).mean() > 0.98, \
        (int(ok.sum()), float(on_surface.mean()), float(occluded.mean()), float(floating.mean()))
    # memory order is spatially coherent (wall after wall, row after row): neighbours in memory are neighbours in space
    step = np.linalg.norm(np.diff(p, axis=0), axis=1)
    assert np.median(step) < 0.05
