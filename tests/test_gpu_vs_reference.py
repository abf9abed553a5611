"""GPU parity against the REFERENCE's own sources, operator tier.

libefusion_hip_nofma.so is the product's HIP code built with -DEF_NO_FMA (every specified fused multiply-add split into
an IEEE multiply and add — the one numerics choice that cannot be read off the reference, see csrc/ef_device.hpp).  In that
build every one of the 16 tracking operators must reproduce, BIT FOR BIT, what the reference's Core/Cuda sources
compute (compiled for the CPU, oracle/_ref): against the committed golden vectors always, and against the live compiled
reference at full resolution when oracle/_ref/libefr_cuda.so travelled with the snapshot.  The product build itself is
compared with the (FMA-specified) oracle in tests/test_gpu_ops_tracking.py.
"""
import os

import numpy as np
import pytest

import efo
import trackops

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tracking_ops_reference.npz")


@pytest.fixture(scope="module")
def nofma_ops():
    from elasticfusion_amd import api, build
    assert os.path.exists(build.NOFMA_LIB), "libefusion_hip_nofma.so not built (python -m elasticfusion_amd.build)"
    api.use_library(build.NOFMA_LIB)
    yield trackops.HipOps(api.ops)
    api.use_library(None)


def test_hip_nofma_reproduces_reference_golden_bits(nofma_ops):
    z = np.load(GOLDEN)
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    ref = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    got = trackops.run_ops(nofma_ops, inp)
    assert set(got) == set(ref)
    for k in ref:
        assert trackops.bits_differ(got[k], ref[k]) == 0, (k, trackops.max_rel(got[k], ref[k]) if ref[k].dtype.kind == "f" else None)


@pytest.mark.parametrize("level", [1, 0])
def test_hip_nofma_equals_live_compiled_reference(nofma_ops, seq, level):
    if not efo.have_reference():
        pytest.skip("oracle/_ref/libefr_cuda.so did not travel with the snapshot (the golden test above covers level 2)")
    f = efo.Fusion()
    for k in range(3):
        rgb, depth, _ = seq.frame(k)
        f.process_frame(rgb, depth, k)
    inp = trackops.make_inputs(f, seq.frame(2)[0], level)
    with efo.backend("reference"):
        ref = trackops.run_ops(efo, inp)
    got = trackops.run_ops(nofma_ops, inp)
    for k in ref:
        assert trackops.bits_differ(got[k], ref[k]) == 0, (level, k)


# ---- map side: the HIP map kernels (no-FMA build) against the reference's own shaders ----
MAP_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "map_passes_reference.npz")


@pytest.fixture(scope="module")
def nofma_map_ops():
    from elasticfusion_amd import api, build
    import mapops
    api.use_library(build.NOFMA_LIB)
    yield mapops.HipMapOps(api)
    api.use_library(None)


def _check_map(got, ref):
    import mapops
    assert set(got) == set(ref)
    for k in ref:
        if k in mapops.INDEX_OUTPUTS:
            continue
        assert got[k].shape == ref[k].shape, (k, got[k].shape, ref[k].shape)
        assert trackops.bits_differ(got[k], ref[k]) == 0, k
    bad, n = mapops.index_pixels_differing(got, ref)
    assert bad <= mapops.INDEX_PIXEL_TOLERANCE * n, (bad, n)   # == 0 since round 5


def test_hip_nofma_map_passes_reproduce_shader_golden_bits(nofma_map_ops):
    import mapops
    z = np.load(MAP_GOLDEN)
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    ref = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    _check_map(mapops.run_passes(nofma_map_ops, inp), ref)


def test_hip_nofma_map_passes_equal_live_compiled_shaders(nofma_map_ops):
    import mapops
    if not efo.have_reference_glsl():
        pytest.skip("oracle/_ref/libefr_glsl.so did not travel with the snapshot (the golden test above covers 96x72)")
    so = efo.reference_glsl_lib()
    so.efg_use_specified_exp(1)
    so.efg_set_depth_compare(1)
    inp = mapops.make_inputs(640, 480)
    with efo.backend("reference_glsl"):
        ref = mapops.run_passes(efo, inp)
    _check_map(mapops.run_passes(nofma_map_ops, inp), ref)


def test_hip_nofma_deformation_equals_compiled_shader(nofma_map_ops):
    """copy_unstable.vert:128-322 (deformation graph) through the compiled shader vs the no-FMA HIP build: bit for bit except
    the lastTime of the handful of surfels whose "seen again" depth lookup sits on a texel edge (N4)."""
    import mapops
    if not efo.have_reference_glsl():
        pytest.skip("oracle/_ref/libefr_glsl.so did not travel with the snapshot")
    so = efo.reference_glsl_lib()
    so.efg_use_specified_exp(1)
    inp = mapops.make_inputs(640, 480)
    graph = mapops.make_graph(inp)
    with efo.backend("reference_glsl"):
        ref = mapops.run_deform(efo, inp, graph)
    got = mapops.run_deform(nofma_map_ops, inp, graph)
    cols = [c for c in range(12) if c != 7]
    assert got.shape == ref.shape and trackops.bits_differ(got[:, cols], ref[:, cols]) == 0
    assert (got[:, 7] != ref[:, 7]).sum() <= 1e-5 * len(got)


# ---- the frame tier's tracker (no-FMA build) against the reference's own tracking driver ----
def test_hip_nofma_tracker_equals_compiled_reference_driver():
    """ef_process_frame's tracker — pyramids, SO(3) pre-alignment, 19 Gauss-Newton iterations with the device-side solve — against
    the reference's own Core/Utils/RGBDOdometry.cpp compiled where it lies (oracle/_ref/libefr_driver.so, over its own CUDA operators):
    fed with the model images and the filtered depth the engine itself produced, the compiled driver must land on the same pose
    and the same six statistics, bit for bit, frame after frame."""
    if not efo.have_reference_driver():
        pytest.skip("oracle/_ref/libefr_driver.so did not travel with the snapshot")
    from elasticfusion_amd import api, build, synth
    W, H = 320, 240
    sq = synth.Sequence(seed=0xEF0005, width=W, height=H)
    api.use_library(build.NOFMA_LIB)
    try:
        ef = api.ElasticFusion(width=W, height=H, fx=sq.fx, fy=sq.fy, cx=sq.cx, cy=sq.cy, maxSurfels=1 << 19)
        frames = [sq.frame(k) for k in range(4)]
        rgba = []
        for rgb, _, _ in frames:
            a = np.full((H, W, 4), 255, np.uint8)
            a[..., :3] = rgb
            rgba.append(a)
        ef.processFrame(frames[0][0], frames[0][1], 0)
        with efo.backend("reference_driver"):
            od = efo.Odometry(W, H, sq.cx, sq.cy, sq.fx, sq.fy)
            od.init_first_rgb(rgba[0])
            for k in (1, 2, 3):
                T_prev = ef.get_T_wc()
                model = [ef.image(n) for n in ("fill_vertex", "fill_normal", "fill_image")]   # young map: the tracker uses the fill-in maps
                ef.processFrame(frames[k][0], frames[k][1], k * 33333)
                st, A, b = ef.trackingStats()
                od.init_icp_model(model[0], model[1], T_prev)
                od.init_rgb_model(model[2])
                od.init_icp(ef.image("depth_filtered"), 20.0)
                od.init_rgb(rgba[k])
                T_ref = od.track(T_prev)
                st_ref, A_ref, b_ref = od.stats()
                assert np.array_equal(np.asarray(st, np.float32).view(np.uint32), st_ref.view(np.uint32)), (k, st, st_ref)
                assert np.array_equal(np.asarray(A).view(np.uint64), A_ref.view(np.uint64)) and np.array_equal(np.asarray(b).view(np.uint64), b_ref.view(np.uint64)), k
                assert np.array_equal(ef.get_T_wc().astype(np.float32), T_ref.astype(np.float32)), (k, np.abs(ef.get_T_wc() - T_ref).max())
        ef.close()
    finally:
        api.use_library(None)
