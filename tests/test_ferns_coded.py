"""The fern database's *_coded entry points (round 3: the codes of a view are computed on the device by k_fern_codes, the host keeps the
inverted-list walk, the view itself is fetched only when needed) replaying the session the REFERENCE's own Core/Ferns.cpp answered in
tests/golden/ferns_reference.npz: with the codes computed HERE in numpy from the fern table — an independent statement of
Ferns.cpp:97-118 on the 1/8 view — every answer must still be the compiled reference's: frames kept, stored codes, matches, recovered
poses, constraints.  And the view must be asked for exactly when Ferns.cpp needs it: when a frame is kept (addFrame) or a keyframe
passes the code gates (findFrame).  Host only: needs neither /root/reference nor a GPU."""
import ctypes as C
import os

import numpy as np

from elasticfusion_amd import api, build
from fernscene import CX, CY, FX, FY, H, W, geometry

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ferns_reference.npz")
VIEW_FETCH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p))
P = C.c_void_p


def codes_of(table, rgb, verts):
    """Ferns.cpp:97-118: fern i tests pixel (x, y) of the view: r, g, b against three byte thresholds, the depth in mm against the fourth"""
    x, y = table[:, 0], table[:, 1]
    z = verts[y, x, 2]
    px = rgb[y, x].astype(np.int64)
    code = ((px[:, 0] > table[:, 2]).astype(np.int64) << 3) | ((px[:, 1] > table[:, 3]).astype(np.int64) << 2) | ((px[:, 2] > table[:, 4]).astype(np.int64) << 1) \
        | ((z * np.float32(1000.0)).astype(np.int32) > table[:, 5]).astype(np.int64)
    good = z > 0
    return np.where(good, code, 255).astype(np.uint8), int(good.sum())


def test_coded_entry_points_answer_like_the_compiled_reference():
    build.build()
    L = api.lib()
    g = np.load(GOLDEN)
    f = api.Ferns(500, 3000, 115.0, W, H, FX, FY, CX, CY, seed=int(g["seed"]))
    table = f.conservatory
    assert np.array_equal(table, g["table"])
    L.ef_ferns_add_frame_coded.argtypes = [P, P, C.c_int, VIEW_FETCH, P, P, C.c_int, C.c_float]
    L.ef_ferns_find_frame_coded.argtypes = [P, P, C.c_int, VIEW_FETCH, P, P, C.c_int, C.c_int, api.FERN_TRACKER, P, P, P, C.c_int, P]
    L.ef_ferns_candidate_possible.argtypes = [P, C.c_int]
    ptr = lambda a: a.ctypes.data_as(P)
    fetches = []

    def fetcher(rgb, verts, norms):
        keep = (np.ascontiguousarray(rgb, np.uint8), np.ascontiguousarray(verts, np.float32), np.ascontiguousarray(norms, np.float32))

        def fn(_user, rgb_out, ch_out, verts_out, norms_out):
            fetches.append(1)
            rgb_out[0], ch_out[0], verts_out[0], norms_out[0] = keep[0].ctypes.data, keep[0].shape[2], keep[1].ctypes.data, keep[2].ctypes.data
            return 0
        return VIEW_FETCH(fn), keep

    kept = []
    for rgb, z, T, t in zip(g["add_rgb"], g["add_z"], g["add_T"], g["add_time"]):
        verts, norms = geometry(z)
        codes, good = codes_of(table, rgb, verts)
        cb, keep = fetcher(rgb, verts, norms)
        before = len(fetches)
        T = np.ascontiguousarray(T, np.float64)
        rc = L.ef_ferns_add_frame_coded(f._h, ptr(codes), good, cb, None, ptr(T), int(t), C.c_float(float(g["threshold"])))
        assert rc in (0, 1), rc
        kept.append(rc)
        assert len(fetches) - before == rc                                    # the view is asked for exactly when the frame is kept
    assert kept == list(g["add_kept"]) and len(f) == len(g["codes"])
    for i in range(len(f)):
        s = f.frame(i)
        assert np.array_equal(s["codes"], g["codes"][i]) and s["goodCodes"] == g["good"][i] and s["srcTime"] == g["src"][i]
    asked = 0
    for rgb, z, (t, lost, err, cnt), delta, closest, Tr, cons, n in zip(g["q_rgb"], g["q_z"], g["q_par"], g["q_delta"], g["q_closest"], g["q_T"], g["q_cons"],
                                                                        g["q_n"]):
        verts, norms = geometry(z)
        codes, good = codes_of(table, rgb, verts)
        cb, keep = fetcher(rgb, verts, norms)
        tracked = []

        def tramp(_user, fv, fn, Tf, cv, cn, Tio, e, k):
            tracked.append(1)
            Te = (np.ctypeslib.as_array(Tio, shape=(16,)).reshape(4, 4).copy() @ delta).reshape(16)
            for i in range(16):
                Tio[i] = Te[i]
            e[0], k[0] = float(err), float(cnt)

        tcb = api.FERN_TRACKER(tramp)
        T_cur = np.ascontiguousarray(g["T_cur"], np.float64)
        T_est = np.zeros((4, 4), np.float64)
        c = np.zeros((500, 6), np.float64)
        nn = C.c_int(0)
        before = len(fetches)
        possible = L.ef_ferns_candidate_possible(f._h, int(t))
        rc = L.ef_ferns_find_frame_coded(f._h, ptr(codes), good, cb, None, ptr(T_cur), int(t), int(bool(lost)), tcb, None, ptr(T_est), ptr(c), 500, C.byref(nn))
        assert rc == closest == f.lastClosest
        assert np.abs(T_est - Tr).max() < 1e-12
        assert nn.value == n and (n == 0 or np.abs(c[:n] - cons[:n]).max() < 1e-12)
        assert len(fetches) - before == len(tracked) <= 1                     # the view is fetched exactly when a keyframe reaches the registration
        assert possible == 1 or (closest == -1 and not tracked)               # "no candidate possible" never contradicts findFrame
        asked += len(tracked)
    assert (g["q_closest"] >= 0).sum() >= 4 and asked >= 4
    f.close()
