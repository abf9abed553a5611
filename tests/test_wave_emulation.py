"""Device code that leans on cross-lane operations, executed WITHOUT a GPU: the text of a kernel is cut out of the product's source and
compiled for the host over tests/wave_emu/hip/hip_runtime.h (execution-space keywords, thread indices, DPP quad permutes with one host
thread per lane).  Used for k_model_maps, whose quad-of-lanes form (coalesced loads, 2x2 boxes through quad_perm exchanges; shipped since
round 3) has to write bit for bit what the one-thread-per-block form of rounds 1-2 wrote (archived as tests/wave_emu/model_maps_block.inc) on maps
with holes, NaNs, both sources (prediction / fill-in) and both modes (world frame / camera frame).  This pins the LOGIC of the variant
(indexing, pairing, the reference's order of additions); what the GPU compiler makes of it is the GPU suite's business."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "elasticfusion_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "wave_emu")


def cut_model_maps(tmp, quad):
    """quad: the product's kernel as shipped; otherwise the product's helpers (argument struct, fill-in choice, planar stores) followed
    by the archived one-thread-per-block kernel of rounds 1-2 (tests/wave_emu/model_maps_block.inc), the form every GPU parity test of
    those rounds validated"""
    src = open(os.path.join(CSRC, "ef_track_kernels.hip")).read()
    a = src.index("// !denseEnough(): float(sum)")
    b = src.index("// ------------------------------------------------------------------------------------------\n// per-pixel Jacobian rows")
    text = src[a:b]
    assert "k_model_maps" in text and "__builtin_amdgcn_mov_dpp" in text and text.count("__global__") == 1
    if not quad:
        text = text[:text.index("// Four lanes share a 4x4 block")] + open(os.path.join(EMU, "model_maps_block.inc")).read()
        assert text.count("__global__") == 1 and "__builtin_amdgcn_mov_dpp" not in text
    path = os.path.join(tmp, "model_maps_cut_%s.inc" % ("quad" if quad else "block"))
    open(path, "w").write(text)
    return path


def build(tmp, quad):
    cut = cut_model_maps(tmp, quad)
    so = os.path.join(tmp, "model_maps_%s.so" % ("quad" if quad else "block"))
    cmd = ["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-I" + EMU, "-I" + CSRC,
           '-DMODEL_MAPS_SOURCE="%s"' % cut] + (["-DEF_MODEL_MAPS_QUAD"] if quad else []) + [os.path.join(EMU, "model_maps_host.cpp"), "-o", so]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    return C.CDLL(so)


def run(lib, pv, pn, fv, fn, cols, rows, camera_frame, R, t, dense):
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    out = [np.full((3 * (rows >> l), cols >> l), -7.0, np.float32) for l in range(3)] + [np.full((3 * (rows >> l), cols >> l), -7.0, np.float32) for l in range(3)]
    depth0 = np.full((rows, cols), -7.0, np.float32)
    lib.run_model_maps.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_int] + [C.c_void_p] * 7
    lib.run_model_maps(P(pv), P(pn), P(fv), P(fn), cols, rows, 6.0, int(camera_frame), P(R), P(t), dense, 100, *[P(o) for o in out], P(depth0))
    return out + [depth0]


@pytest.mark.parametrize("cols,rows", [(80, 60), (132, 52)])
def test_quad_of_lanes_model_maps_equals_the_validated_kernel(tmp_path, cols, rows):
    block, quad = build(str(tmp_path), False), build(str(tmp_path), True)
    rng = np.random.default_rng(cols)

    def maps(seed):
        g = np.random.default_rng(seed)
        v = g.normal(0, 1, (rows, cols, 4)).astype(np.float32)
        v[..., 2] = np.abs(v[..., 2]) + 0.5
        n = g.normal(0, 1, (rows, cols, 4)).astype(np.float32)
        hole = g.random((rows, cols)) < 0.15
        v[hole, 2] = 0                                           # empty texels
        v[g.random((rows, cols)) < 0.05, 0] = np.nan             # NaN in x only
        n[g.random((rows, cols)) < 0.05, 0] = np.nan
        v[g.random((rows, cols)) < 0.03, 2] = 7.5                # beyond maxDepthRGB
        v[8:16, 8:24, 2] = 0                                     # a whole region of empty 4x4 blocks
        return v, n

    pv, pn = maps(1)
    fv, fn = maps(2)
    a = 0.3
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32).reshape(9)
    t = np.array([0.1, -0.2, 0.3], np.float32)
    for camera_frame in (False, True):
        for dense in (10, 90):                                   # fill-in maps / predicted maps (denseEnough)
            want = run(block, pv, pn, fv, fn, cols, rows, camera_frame, R, t, dense)
            got = run(quad, pv, pn, fv, fn, cols, rows, camera_frame, R, t, dense)
            for k, (w, g) in enumerate(zip(want, got)):
                assert np.array_equal(w.view(np.uint32), g.view(np.uint32)), (camera_frame, dense, k, int((w.view(np.uint32) != g.view(np.uint32)).sum()))
            assert np.isnan(want[0]).any() and (want[2] != -7.0).any()
            # quirk Q3: the resized levels only get an x-plane NaN; y/z of an invalid texel keep what was there (the -7 canary)
            lvl1 = want[1]
            assert (lvl1[rows // 2:] == -7.0).any()


def build_solve(tmp):
    so = os.path.join(tmp, "solve_shipped.so")
    cmd = ["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-I" + EMU, "-I" + CSRC, os.path.join(EMU, "solve_host.cpp"), "-o", so]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    return C.CDLL(so)


def test_wave_parallel_ldlt_equals_the_scalar_statement(tmp_path):
    """the 6x6 LDL^T of the update step as the tracker runs it (one matrix element per lane, 64 emulated lanes, diagonal broadcasts
    through v_readlane) against the scalar restatement of Eigen::LDLT it mirrors (which tests/test_oracle_linalg.py and the GPU operator
    tests tie to the oracle), on well-conditioned, pivoting-heavy and degenerate systems"""
    shipped = build_solve(str(tmp_path))
    rng = np.random.RandomState(0)
    cases = []
    for t in range(12):
        J = rng.randn(40, 6) * np.array([100, 100, 100, 1, 1, 1.0])
        A = J.T @ J
        if t % 4 == 3:
            A[2, :] = 0; A[:, 2] = 0                            # a singular direction
        if t % 4 == 2:
            A = A[::-1, ::-1].copy()                            # largest pivots last: every step swaps
        cases.append((A, rng.randn(6)))
    cases.append((np.zeros((6, 6)), rng.randn(6)))              # a frame without correspondences
    v = rng.randn(6)
    cases.append((np.outer(v, v), rng.randn(6)))                # rank one
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    for k, (A, b) in enumerate(cases):
        A = np.ascontiguousarray((A + A.T) / 2 if k < 12 else A, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        want, got = np.zeros(6), np.zeros(6)
        shipped.run_ldlt6_scalar(P(A), P(b), P(want))
        shipped.run_ldlt6_wave(P(A), P(b), P(got))
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (k, got, want)
        got2 = np.zeros(6)
        shipped.run_ldlt6_every_lane(P(A), P(b), P(got2))     # the whole-matrix-per-lane variant (-DEF_LDLT_EVERY_LANE)
        assert np.array_equal(got2.view(np.uint64), want.view(np.uint64)), (k, got2, want)
