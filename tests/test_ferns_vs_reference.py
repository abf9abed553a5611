"""The product's fern database (include/ef_hip.h ef_ferns_*, elasticfusion_amd/csrc/ef_ferns.hip) and the oracle's restatement
(oracle/efo_ferns.cpp) against the reference's own Core/Ferns.cpp, compiled from /root/reference where it lies inside oracle/_ref/libefr_frame.so (oracle/Makefile `refframe`): the
bridge queues the three Resize read-backs of every addFrame / findFrame from the same arrays the product gets, and the 80x60
tracker inside findFrame is a scripted double on both sides (same pose increment, same ICP statistics).  Compared: the fern table
drawn from a seed (generateFerns itself), which frames are kept, their codes, which stored frame a view is matched to, the
recovered pose, the surface constraints, and the two private measures (blockHDAware, photometricCheck)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from elasticfusion_amd import build
from elasticfusion_amd import api
import efo

BACKENDS = {"product": api.Ferns, "oracle": efo.Ferns}

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libefr_frame.so")
from fernscene import CX, CY, FX, FY, H, W, h, place, pose, w   # noqa: E402

P = C.c_void_p


def have():
    if os.path.isdir("/root/reference/Core"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refframe"])
    return os.path.exists(SO)


pytestmark = pytest.mark.skipif(not have(), reason="oracle/_ref/libefr_frame.so can only be built where /root/reference exists")


@pytest.fixture(scope="module")
def ref():
    build.build()
    so = C.CDLL(SO)
    so.efe_create.restype = P
    so.efe_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float,
                              C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_char_p]
    so.efe_destroy.argtypes = [P]
    so.efe_script_readbacks.argtypes = [C.c_int, C.c_uint, P, C.c_long]
    so.efe_script_tracker.argtypes = [P, C.c_float, C.c_float, C.c_double, C.c_int]
    so.efe_ferns_num.argtypes = [P]
    so.efe_ferns_reseed.argtypes = [P, C.c_uint, P]
    so.efe_ferns_add_frame.argtypes = [P, P, P, P, P, C.c_int, C.c_float]
    so.efe_ferns_find_frame.argtypes = [P, P, P, P, P, C.c_int, C.c_int, P, P, C.c_int, P]
    so.efe_ferns_count.argtypes = [P]
    so.efe_ferns_frame.argtypes = [P, C.c_int, P, P, P, P]
    so.efe_ferns_block_hd_aware.argtypes = [P, C.c_int, C.c_int]
    so.efe_ferns_block_hd_aware.restype = C.c_float
    so.efe_ferns_photometric_check.argtypes = [P, P, P, P, C.c_int]
    so.efe_ferns_photometric_check.restype = C.c_float
    so.efe_script_readbacks(-1, 0, None, 0)
    handles = []

    def make(tmp):
        # ElasticFusion(depthCut = 3) builds Ferns(500, 3000, photoThresh = 115) (ElasticFusion.cpp:53)
        hd = P(so.efe_create(W, H, FX, FY, CX, CY, 200, 35000, 5e-5, 1e-5, 1, 10.0, 3.0, 10.0, 0, 1, 0, os.path.join(tmp, "ferns").encode()))
        handles.append(hd)
        return hd

    so.make = make
    yield so
    for hd in handles:
        so.efe_destroy(hd)


def both_add(so, hd, f, view, T, t, thresh):
    rgb, verts, norms = view
    T = np.ascontiguousarray(T, np.float64)
    a = so.efe_ferns_add_frame(hd, rgb.ctypes.data, verts.ctypes.data, norms.ctypes.data, T.ctypes.data, t, thresh)
    b = f.addFrame(rgb, verts, norms, T, t, thresh)
    assert bool(a) == b, (t, a, b)
    return b


def both_find(so, hd, f, view, T, t, lost, delta, err, cnt):
    rgb, verts, norms = view
    T = np.ascontiguousarray(T, np.float64)
    d = np.ascontiguousarray(delta, np.float64)
    so.efe_script_tracker(d.ctypes.data, err, cnt, 1e-7, 0)
    Tr = np.zeros((4, 4))
    cr = np.zeros((600, 6))
    n = C.c_int(0)
    closest = so.efe_ferns_find_frame(hd, rgb.ctypes.data, verts.ctypes.data, norms.ctypes.data, T.ctypes.data, t, int(lost), Tr.ctypes.data, cr.ctypes.data, 600,
                                      C.byref(n))
    seen = {}

    def tracker(fv, fn, Tf, cv, cn, Tin):
        seen["fern"] = (fv.copy(), fn.copy(), Tf.copy())
        assert np.array_equal(cv, verts) and np.array_equal(cn, norms) and np.array_equal(Tin, Tf)
        return Tin @ delta, err, cnt

    Tp, cp = f.findFrame(rgb, verts, norms, T, t, lost, tracker)
    assert f.lastClosest == closest, (f.lastClosest, closest)
    assert np.abs(Tp - Tr).max() < 1e-12
    assert len(cp) == n.value
    if n.value:
        assert np.abs(cp - cr[:n.value]).max() < 1e-12
    return closest, Tp, cp, seen


@pytest.mark.parametrize("backend", list(BACKENDS))
def test_fern_table_from_seed_is_the_references(ref, tmp_path, backend):
    Ferns = BACKENDS[backend]
    hd = ref.make(str(tmp_path))
    num = ref.efe_ferns_num(hd)
    assert num == 500
    for seed in (0, 1234, 2**31 + 5):
        t = np.zeros((num, 6), np.int32)
        ref.efe_ferns_reseed(hd, seed, t.ctypes.data)
        f = Ferns(num, 3000, 115.0, W, H, FX, FY, CX, CY, seed=seed)
        assert np.array_equal(f.conservatory, t)
        assert t[:, 0].max() < w and t[:, 1].max() < h and t[:, 5].min() >= 400 and t[:, 5].max() <= 3000
        f.close()


@pytest.mark.parametrize("backend", list(BACKENDS))
def test_keyframe_selection_matching_and_constraints(ref, tmp_path, backend):
    Ferns = BACKENDS[backend]
    hd = ref.make(str(tmp_path))
    num = ref.efe_ferns_num(hd)
    table = np.zeros((num, 6), np.int32)
    ref.efe_ferns_reseed(hd, 99, table.ctypes.data)
    f = Ferns(num, 3000, 115.0, W, H, FX, FY, CX, CY, seed=99)
    thresh = 0.3095   # MainController's fernThresh
    # 8 places, 5 jittered views each: the first view of a place is new, its repeats are too similar
    kept, tick, poses = [], 0, {}
    for k in range(8):
        T = pose([0.2, 1, 0.1], 0.4 * k, [0.3 * k, 0.05 * k, -0.1 * k])
        for j in range(5):
            if both_add(ref, hd, f, place(k, jitter=j), T, tick, thresh):
                kept.append((k, j))
                poses[len(kept) - 1] = T
            tick += 1
    assert len(f) == ref.efe_ferns_count(hd) == len(kept)
    assert [k for k, _ in kept] == sorted(set(k for k, _ in kept)) and len(kept) >= 6   # one frame per distinguishable place
    # a view without any depth is never stored
    empty = place(3)
    empty[1][...] = 0
    assert not both_add(ref, hd, f, empty, np.eye(4), tick, thresh)
    for i in range(len(kept)):
        codes = np.zeros(num, np.uint8)
        good, src = C.c_int(0), C.c_int(0)
        T = np.zeros((4, 4))
        ref.efe_ferns_frame(hd, i, codes.ctypes.data, C.byref(good), C.byref(src), T.ctypes.data)
        s = f.frame(i)
        assert np.array_equal(s["codes"], codes) and s["goodCodes"] == good.value and s["srcTime"] == src.value
        assert np.abs(s["T_wc"] - T).max() < 1e-15 and np.abs(T - poses[i]).max() < 1e-12
        assert (codes == 255).sum() == num - good.value and good.value < num   # the hole leaves bad codes
        k, j = kept[i]
        assert np.array_equal(s["rgb"], place(k, jitter=j)[0]) and np.array_equal(s["verts"], place(k, jitter=j)[1])
    for a in range(len(kept)):
        for b in range(len(kept)):
            assert f.blockHDAware(a, b) == ref.efe_ferns_block_hd_aware(hd, a, b)
        assert f.blockHDAware(a, a) == 1.0

    small = pose([1, 0.3, 0.2], 0.004, [0.003, -0.002, 0.004])
    now = tick + 400
    T_cur = pose([0, 1, 0], 0.1, [1.0, 0.2, 0.3])
    # a revisit of place 2, tracker converges: matched, pose recovered, constraints made
    closest, T_est, cons, seen = both_find(ref, hd, f, place(2, jitter=9), T_cur, now, False, small, 1e-4, 4000.0)
    i2 = [k for k, _ in kept].index(2)
    assert closest == i2 and len(cons) > 30
    assert np.abs(T_est - poses[i2] @ small).max() < 1e-12
    assert np.array_equal(seen["fern"][0], f.frame(i2)["verts"]) and np.abs(seen["fern"][2] - poses[i2]).max() < 1e-15
    v = place(2, jitter=9)[1]
    tab = f.conservatory
    used = [i for i in range(0, num, num // 50) if v[tab[i, 1], tab[i, 0], 2] > 0 and int(np.float32(v[tab[i, 1], tab[i, 0], 2]) * np.float32(1000)) < 3000]
    assert len(cons) == len(used)
    p = np.concatenate([v[tab[used[0], 1], tab[used[0], 0], :3].astype(np.float64), [1.0]])
    assert np.abs(cons[0, :3] - (T_cur @ p)[:3]).max() < 1e-12 and np.abs(cons[0, 3:] - (T_est @ p)[:3]).max() < 1e-12
    # the gates: ICP error, ICP count (2400 tracking / 1400 lost), the photometric check, and the age of the stored frame
    outcomes = [both_find(ref, hd, f, place(2, jitter=11), T_cur, now, lost, small, 1e-4, cnt)[0] for lost, cnt in [(False, 2400.0), (True, 2400.0), (True, 1400.0)]]
    assert outcomes == [-1, i2, -1]
    assert both_find(ref, hd, f, place(2, jitter=11), T_cur, now, False, pose([0, 1, 0], 0.0, [0.3, 0.3, 0]), 1e-4, 4000.0)[0] == -1   # photometric gate (115)
    assert both_find(ref, hd, f, place(2, jitter=11), T_cur, now, False, small, 4e-4, 4000.0)[0] == -1     # ICP error gate (3e-4)
    src2 = f.frame(i2)["srcTime"]
    assert both_find(ref, hd, f, place(2, jitter=11), T_cur, src2 + 300, False, small, 1e-4, 4000.0)[0] != i2   # stored too recently
    assert both_find(ref, hd, f, place(2, jitter=11), T_cur, src2 + 301, False, small, 1e-4, 4000.0)[0] == i2
    # every place revisited, and a place never seen
    for k in range(8):
        both_find(ref, hd, f, place(k, jitter=13), T_cur, now, k % 2 == 1, small, 2e-4, 3000.0)
    both_find(ref, hd, f, place(31), T_cur, now, False, small, 1e-4, 4000.0)
    # photometricCheck directly, on registered and mis-registered poses
    for k, d in [(2, np.eye(4)), (2, small), (5, small), (2, pose([0, 1, 0], 0.2, [0.1, 0, 0]))]:
        i = [kk for kk, _ in kept].index(k) if k in [kk for kk, _ in kept] else 0
        rgb, verts, _ = place(k, jitter=3)
        Te = np.ascontiguousarray(poses[i] @ d)
        a = ref.efe_ferns_photometric_check(hd, rgb.ctypes.data, verts.ctypes.data, Te.ctypes.data, i)
        b = f.photometricCheck(rgb, verts, Te, i)
        assert a == b or (np.isnan(a) and np.isnan(b)), (k, a, b)
    rgb, verts, _ = place(2, jitter=0)
    assert f.photometricCheck(rgb, verts, poses[i2], i2) < 8.0     # the stored view against itself (the projection truncates, so a pixel may shift by one)
    f.close()


def test_arguments_and_state():
    build.build()
    Ferns = api.Ferns
    f = Ferns(500, 3000, 115.0, W, H, FX, FY, CX, CY, seed=3)
    rgb, verts, norms = place(0)
    t = f.conservatory
    f.conservatory = t[::-1].copy()
    assert np.array_equal(f.conservatory, t[::-1])
    assert f.addFrame(np.dstack([rgb, np.full((h, w), 255, np.uint8)]), verts, norms, np.eye(4), 0, 0.3)   # RGBA rows are accepted
    assert np.array_equal(f.frame(0)["rgb"], rgb)
    with pytest.raises(Exception):
        f.conservatory = t       # the table is frozen once a frame refers to it
    T_est, cons = f.findFrame(rgb, verts, norms, np.eye(4), 100, False, lambda *a: (_ for _ in ()).throw(AssertionError("not called")))
    assert f.lastClosest == -1 and len(cons) == 0 and np.array_equal(T_est, np.eye(4))   # too recent: no candidate, tracker not run
    f.setFramePose(0, pose([0, 0, 1], 0.3, [1, 2, 3]))
    assert np.abs(f.frame(0)["T_wc"] - pose([0, 0, 1], 0.3, [1, 2, 3])).max() < 1e-15
    f.close()
