#!/usr/bin/env python
"""Generates tests/golden/ferns_reference.npz from the REFERENCE's own Core/Ferns.cpp compiled for the CPU (oracle/_ref/libefr_frame.so,
`make -C oracle refframe`; oracle/ref_frame_bridge.cpp efe_ferns_*): a scripted session — the fern table drawn by generateFerns from
seed 20260922, 12 addFrame calls (4 places x 3 noisy views), 10 findFrame calls with a scripted tracker — with every input (colour
+ depth of each 80x60 view, poses, times, tracker answers) and every answer of the reference (kept or not, the stored codes, the
matched frame, the recovered pose, the constraints).  Must be run where /root/reference exists; tests/test_ferns_golden.py then
replays the session on the product's ef_ferns_* anywhere.

    python tools/make_ferns_golden.py
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fernscene import CX, CY, FX, FY, H, W, geometry, place, pose  # noqa: E402

P = C.c_void_p
SEED = 20260922


def main():
    so = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libefr_frame.so"))
    so.efe_create.restype = P
    so.efe_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float,
                              C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_char_p]
    so.efe_script_readbacks.argtypes = [C.c_int, C.c_uint, P, C.c_long]
    so.efe_script_tracker.argtypes = [P, C.c_float, C.c_float, C.c_double, C.c_int]
    so.efe_ferns_reseed.argtypes = [P, C.c_uint, P]
    so.efe_ferns_add_frame.argtypes = [P, P, P, P, P, C.c_int, C.c_float]
    so.efe_ferns_find_frame.argtypes = [P, P, P, P, P, C.c_int, C.c_int, P, P, C.c_int, P]
    so.efe_ferns_frame.argtypes = [P, C.c_int, P, P, P, P]
    so.efe_ferns_count.argtypes = [P]
    so.efe_script_readbacks(-1, 0, None, 0)
    tmp = tempfile.mkdtemp()
    hd = P(so.efe_create(W, H, FX, FY, CX, CY, 200, 35000, 5e-5, 1e-5, 1, 10.0, 3.0, 10.0, 0, 1, 0, os.path.join(tmp, "ferns").encode()))
    table = np.zeros((500, 6), np.int32)
    so.efe_ferns_reseed(hd, SEED, table.ctypes.data)

    g = {"seed": np.array(SEED), "table": table, "threshold": np.float32(0.3095)}
    add_rgb, add_z, add_T, add_time, add_kept = [], [], [], [], []
    tick = 0
    for k in range(4):
        T = pose([0.2, 1, 0.1], 0.4 * k, [0.3 * k, 0.05 * k, -0.1 * k])
        for j in range(3):
            rgb, verts, norms = place(k, jitter=j)
            kept = so.efe_ferns_add_frame(hd, rgb.ctypes.data, verts.ctypes.data, norms.ctypes.data, T.ctypes.data, tick, 0.3095)
            add_rgb.append(rgb); add_z.append(verts[..., 2].copy()); add_T.append(T); add_time.append(tick); add_kept.append(kept)
            tick += 7
    n = so.efe_ferns_count(hd)
    codes = np.zeros((n, 500), np.uint8)
    good = np.zeros(n, np.int32)
    src = np.zeros(n, np.int32)
    for i in range(n):
        a, b, T = C.c_int(0), C.c_int(0), np.zeros(16)
        so.efe_ferns_frame(hd, i, codes[i].ctypes.data, C.byref(a), C.byref(b), T.ctypes.data)
        good[i], src[i] = a.value, b.value
    g.update(add_rgb=np.stack(add_rgb), add_z=np.stack(add_z), add_T=np.stack(add_T), add_time=np.array(add_time, np.int32),
             add_kept=np.array(add_kept, np.int32), codes=codes, good=good, src=src)

    small = pose([1, 0.3, 0.2], 0.004, [0.003, -0.002, 0.004])
    far = pose([0, 1, 0], 0.0, [0.3, 0.3, 0])
    T_cur = pose([0, 1, 0], 0.1, [1.0, 0.2, 0.3])
    # place, jitter, time, lost, tracker increment, ICP error, ICP count
    queries = [(0, 5, 600, 0, small, 1e-4, 4000.0), (1, 5, 600, 0, small, 1e-4, 4000.0), (2, 5, 600, 1, small, 2e-4, 2000.0), (3, 5, 600, 0, small, 1e-4, 2400.0),
               (2, 6, 600, 0, far, 1e-4, 4000.0), (1, 6, 600, 0, small, 4e-4, 4000.0), (0, 6, 300, 0, small, 1e-4, 4000.0), (0, 6, 301, 0, small, 1e-4, 4000.0),
               (9, 0, 600, 0, small, 1e-4, 4000.0), (3, 6, 600, 1, small, 1e-4, 1401.0)]
    q_rgb, q_z, q_par, q_delta, q_closest, q_T, q_cons, q_n = [], [], [], [], [], [], [], []
    for k, j, t, lost, delta, err, cnt in queries:
        rgb, verts, norms = place(k, jitter=j)
        d = np.ascontiguousarray(delta)
        so.efe_script_tracker(d.ctypes.data, err, cnt, 1e-7, 0)
        Te, cons, m = np.zeros((4, 4)), np.zeros((64, 6)), C.c_int(0)
        closest = so.efe_ferns_find_frame(hd, rgb.ctypes.data, verts.ctypes.data, norms.ctypes.data, T_cur.ctypes.data, t, lost, Te.ctypes.data, cons.ctypes.data,
                                          64, C.byref(m))
        q_rgb.append(rgb); q_z.append(verts[..., 2].copy()); q_par.append([t, lost, err, cnt]); q_delta.append(d); q_closest.append(closest)
        q_T.append(Te); q_cons.append(cons); q_n.append(m.value)
        assert np.array_equal(geometry(verts[..., 2])[0], verts)
    g.update(T_cur=T_cur, q_rgb=np.stack(q_rgb), q_z=np.stack(q_z), q_par=np.array(q_par, np.float64), q_delta=np.stack(q_delta),
             q_closest=np.array(q_closest, np.int32), q_T=np.stack(q_T), q_cons=np.stack(q_cons), q_n=np.array(q_n, np.int32))
    path = os.path.join(ROOT, "tests", "golden", "ferns_reference.npz")
    np.savez_compressed(path, **g)
    print(path, os.path.getsize(path), "bytes; kept", add_kept, "closest", q_closest, "constraints", q_n)


if __name__ == "__main__":
    main()
