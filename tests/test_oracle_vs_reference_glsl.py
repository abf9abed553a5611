"""Pins the MAP side of the oracle against the REFERENCE's own shaders, on the CPU.

oracle/_ref/libefr_glsl.so = the 18 hot-path shaders of Core/Shaders (depth_bilateral, depth_metric, vertex_feedback.{vert,geom},
init_unstable, index_map.{vert,frag}, splat.vert, combo_splat.frag, fill_{vertex,normal,rgb}.frag, data.{vert,geom,frag},
update.vert, copy_unstable.{vert,geom} + surfels/geometry/color.glsl) compiled where they lie by g++ through
oracle/glsl_on_cpu/ (GLSL vocabulary in C++; -ffp-contract=off), driven pass by pass by oracle/ref_glsl_bridge.cpp with the
reference's uniforms and the fixed-function stages as specified (N1-N5).  Claim: the oracle built without fused
multiply-adds reproduces every pass BIT FOR BIT — bilateral filter (with the specified exp), metric depth, first-frame
seeding, the surfel splat with its ray/disc intersection, fill-in, the 16-tap association + merge (data.vert, update.vert)
and the clean pass with its float tap loops (copy_unstable.vert) — except for the index map, where index_map.vert's round
trip through NDC moves a handful of points across a pixel edge (mapops.INDEX_PIXEL_TOLERANCE).
"""
import ctypes as C

import numpy as np
import pytest

import efo
import mapops
import trackops

pytestmark = pytest.mark.skipif(not efo.have_reference_glsl(), reason="oracle/_ref/libefr_glsl.so absent and /root/reference not present to build it")


@pytest.fixture(scope="module")
def inputs():
    return mapops.make_inputs(640, 480)


def test_map_passes_against_compiled_shaders(inputs):
    so = efo.reference_glsl_lib()
    so.efg_use_specified_exp(1)     # exp() = the oracle's IEEE-only polynomial: exp-dependent outputs comparable bit for bit
    so.efg_set_depth_compare(1)     # N2 as specified: depth test on the camera-space z
    with efo.backend("reference_glsl"):
        ref = mapops.run_passes(efo, inputs)
    with efo.backend("nofma"):
        got = mapops.run_passes(efo, inputs)
    spec = mapops.run_passes(efo, inputs)
    # the fixture exercises the interesting paths
    assert (ref["fuse_new_unstable"][:, 7] == -1).sum() > 1000 and (ref["fuse_new_unstable"][:, 7] == -2).sum() > 10
    assert len(ref["clean_map"]) != len(inputs["s2"]) + len(inputs["nu"]) and len(ref["seed_map"]) > 200000
    assert (ref["predict_vertex"][..., 2] > 0).sum() > 50000
    for k in ref:
        if k in mapops.INDEX_OUTPUTS:
            continue
        assert got[k].shape == ref[k].shape, (k, got[k].shape, ref[k].shape)
        assert trackops.bits_differ(got[k], ref[k]) == 0, k
    bad, n = mapops.index_pixels_differing(got, ref)
    assert bad <= mapops.INDEX_PIXEL_TOLERANCE * n, (bad, n)
    # the FMA-specified oracle: same surfel counts / ids / integer images, floats within FMA rounding
    for k in ("filter_depth", "predict_time", "fill_image", "passthrough_image"):
        assert trackops.bits_differ(spec[k], ref[k]) == 0, k
    for k in ("seed_map", "fuse_map", "fuse_new_unstable", "clean_map"):
        assert spec[k].shape == ref[k].shape, k
        assert trackops.max_rel(spec[k][:, [0, 1, 2, 3, 8, 9, 10, 11]], ref[k][:, [0, 1, 2, 3, 8, 9, 10, 11]]) <= 1e-5, k
        assert np.array_equal(spec[k][:, 4:8], ref[k][:, 4:8]), k     # packed colour, init/last time, tags


def test_depth_buffer_rule_is_the_only_difference_in_the_splat(inputs):
    """With the depth test on what the shader writes (gl_FragDepth = z / 40 + 0.5 in float32: ~2.4 um resolution) instead
    of on z, fragments of neighbouring surfels on one surface tie and the earlier draw wins: a fraction of a percent of
    the pixels show another surfel.  N2 (compare z) is the specification the oracle and the HIP kernels follow."""
    so = efo.reference_glsl_lib()
    so.efg_use_specified_exp(1)
    c = inputs["cam"]
    cam = efo.make_cam(int(c[0]), int(c[1]), *[float(x) for x in c[2:]])
    T, tick = inputs["T"].reshape(4, 4), int(inputs["tick"].reshape(-1)[0])
    outs = []
    for mode in (1, 0):
        so.efg_set_depth_compare(mode)
        with efo.backend("reference_glsl"):
            outs.append(efo.combined_predict(cam, T, inputs["surf"], mapops.MAXD, mapops.CONF, tick, tick, mapops.TD))
    so.efg_set_depth_compare(1)
    differing = (outs[0][1].view(np.uint32) != outs[1][1].view(np.uint32)).any(axis=-1).mean()
    assert 0 < differing < 0.02, differing


def test_inactive_prediction_against_compiled_shader(inputs):
    """IndexMap::combinedPredict(..., INACTIVE) as the local loop closure calls it (ElasticFusion.cpp:451-459): time = 0,
    maxTime = tick - timeDelta, so only surfels NOT seen inside the window are drawn.  No-FMA oracle vs splat.vert +
    combo_splat.frag: bit for bit, and the view is a proper subset of the ACTIVE one."""
    so = efo.reference_glsl_lib()
    so.efg_use_specified_exp(1)
    so.efg_set_depth_compare(1)
    c = inputs["cam"]
    cam = efo.make_cam(int(c[0]), int(c[1]), *[float(x) for x in c[2:]])
    T, tick = inputs["T"].reshape(4, 4), int(inputs["tick"].reshape(-1)[0])
    with efo.backend("reference_glsl"):
        ref = efo.combined_predict(cam, T, inputs["surf"], mapops.MAXD, mapops.CONF, 0, tick - 2, 2)
        act = efo.combined_predict(cam, T, inputs["surf"], mapops.MAXD, mapops.CONF, tick, tick, mapops.TD)
    with efo.backend("nofma"):
        got = efo.combined_predict(cam, T, inputs["surf"], mapops.MAXD, mapops.CONF, 0, tick - 2, 2)
    for a, b in zip(got, ref):
        assert trackops.bits_differ(a, b) == 0
    n_old, n_act = int((ref[1][..., 2] > 0).sum()), int((act[1][..., 2] > 0).sum())
    assert 1000 < n_old < n_act, (n_old, n_act)
    assert ref[3][ref[1][..., 2] > 0].max() <= tick - 2


def test_graph_sampling_against_compiled_shader(inputs):
    """SURVEY §8f row 4, device part: Deformation::sampleGraphModel = sample.vert + sample.geom under transform feedback"""
    surf = inputs["surf"]
    with efo.backend("reference_glsl"):
        ref = efo.sample_graph(surf)
    got = efo.sample_graph(surf)
    assert len(ref) == (len(surf) - 1) // 5000 + 1 and trackops.bits_differ(got, ref) == 0
    assert np.array_equal(ref[:, :3], surf[::5000, :3]) and np.array_equal(ref[:, 3], surf[::5000, 6])


@pytest.mark.parametrize("factor", [20, 8])
def test_resize_against_compiled_shader(inputs, factor):
    """Resize::vertex / Resize::image = empty.vert + quad.geom + resize.frag: the constraint grid (factor 20) and the fern
    database's inputs (factor 8) — the sample point falls exactly on a texel boundary and takes the upper texel (G8 / N4)."""
    vt, img = inputs["vt"], inputs["img"]
    for a in (vt, img):
        with efo.backend("reference_glsl"):
            ref = efo.resize_nearest(a, factor)
        got = efo.resize_nearest(a, factor)
        assert ref.shape == (a.shape[0] // factor, a.shape[1] // factor, 4)
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
        assert np.array_equal(got, a[factor // 2::factor, factor // 2::factor][:ref.shape[0], :ref.shape[1]])
    assert (efo.resize_nearest(vt, factor)[..., 2] > 0).sum() > 10


def test_bilateral_with_libm_exp(inputs):
    """exp() through libm's expf instead of the specified polynomial: the filtered depth may differ by at most 1 mm, rarely"""
    so = efo.reference_glsl_lib()
    raw = inputs["raw"][100:180, 200:360].copy()
    so.efg_use_specified_exp(0)
    with efo.backend("reference_glsl"):
        a = efo.filter_depth(raw, 3.0)
    so.efg_use_specified_exp(1)
    b = efo.filter_depth(raw, 3.0)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.01, (d.max(), (d > 0).mean())


def test_deformation_graph_application_against_compiled_shader(inputs):
    """copy_unstable.vert:128-322 — binary search of the node by time, 20 nearest-in-time candidates,
    the shader's exchange sort, k = 4 blend weights, rigid blend of position and normal, the "seen again" test against the
    synthesized depth.  No-FMA oracle vs the compiled shader: bit for bit, and the graph really moves the map."""
    so = efo.reference_glsl_lib()
    so.efg_use_specified_exp(1)
    graph = mapops.make_graph(inputs)
    with efo.backend("reference_glsl"):
        ref = mapops.run_deform(efo, inputs, graph)
        ref_fern = mapops.run_deform(efo, inputs, graph, isFern=1)
    with efo.backend("nofma"):
        got = mapops.run_deform(efo, inputs, graph)
        got_fern = mapops.run_deform(efo, inputs, graph, isFern=1)
        undeformed = efo.clean(efo.make_cam(*[int(x) for x in inputs["cam"][:2]], *[float(x) for x in inputs["cam"][2:]]),
                               inputs["T"].reshape(4, 4), int(inputs["tick"].reshape(-1)[0]), inputs["idx2"], inputs["vc2"], inputs["ct2"],
                               inputs["nr2"], mapops.CONF, mapops.TD, mapops.MAXD, inputs["s2"], inputs["nu"])
    assert got.shape == ref.shape == undeformed.shape
    assert trackops.bits_differ(got_fern, ref_fern) == 0
    # the only data-dependent NEAREST lookup at an arbitrary coordinate is the "seen again" depth test
    # (textureLod(depthSampler, vec2(x / cols, y / rows))): x / cols * cols can land on the other side of a texel edge than
    # floor(x) (N4) -> lastTime of at most a handful of surfels; everything else bit for bit
    cols_but_last_time = [c for c in range(12) if c != 7]
    assert trackops.bits_differ(got[:, cols_but_last_time], ref[:, cols_but_last_time]) == 0
    assert (got[:, 7] != ref[:, 7]).sum() <= 1e-5 * len(got)
    moved = np.linalg.norm(got[:, :3] - undeformed[:, :3], axis=1)
    assert (moved > 1e-3).mean() > 0.9 and moved.max() < 0.5          # everything but this frame's points was deformed
    assert (got[:, 7] != got_fern[:, 7]).any()                         # the depth test updates lastTime only when !isFern
    spec = mapops.run_deform(efo, inputs, graph)
    assert trackops.max_rel(spec[:, :3], ref[:, :3]) <= 1e-5 and np.array_equal(spec[:, 4:7], ref[:, 4:7])
    assert (spec[:, 7] != ref[:, 7]).sum() <= 1e-5 * len(spec)
