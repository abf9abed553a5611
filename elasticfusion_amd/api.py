"""Host-side mirror of the reference interface over the C ABI of libefusion_hip.so (include/ef_hip.h).

``ElasticFusion`` keeps the reference's public names (Core/ElasticFusion.h:40-255): processFrame, predict,
get_T_wc, getTick/setTick, getTimeDelta, getConfidenceThreshold, setRgbOnly/..., savePly; the ctor takes the
reference's argument names plus the Resolution / Intrinsics singletons as plain arguments.
``ops`` exposes the operator tier (the cudafuncs.cuh free functions and the GLSL passes) on numpy arrays:
each call uploads, runs the HIP kernel(s) through the C ABI and downloads — it is what the parity tests use.

There is NO fallback: if the shared library is missing or no GPU is present, everything here raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# EF_HIP_LIB: developer override (A/B of two builds on one GPU box, tools/ab_bench.py); never a CPU fallback
LIB_PATH = os.environ.get("EF_HIP_LIB") or os.path.join(_HERE, "libefusion_hip.so")
_lib = None

c_f, c_i, c_u32, P = C.c_float, C.c_int, C.c_uint32, C.c_void_p


class EFError(RuntimeError):
    pass


class ef_config(C.Structure):
    _fields_ = [("width", c_i), ("height", c_i), ("fx", c_f), ("fy", c_f), ("cx", c_f), ("cy", c_f),
                ("time_delta", c_i), ("confidence", c_f), ("depth_cut", c_f), ("icp_weight", c_f),
                ("fast_odom", c_i), ("so3", c_i), ("frame_to_frame_rgb", c_i), ("pyramid", c_i), ("rgb_only", c_i),
                ("close_loops", c_i), ("max_surfels", c_u32), ("device", c_i), ("stream", P)]


class ef_intr(C.Structure):
    _fields_ = [("fx", c_f), ("fy", c_f), ("cx", c_f), ("cy", c_f)]


class ef_cam(C.Structure):
    _fields_ = [("cols", c_i), ("rows", c_i), ("fx", c_f), ("fy", c_f), ("cx", c_f), ("cy", c_f)]


class ef_timing(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", c_f)]


DATATERM = np.dtype([("zero", np.int16, 2), ("one", np.int16, 2), ("diff", np.float32), ("valid", np.uint8),
                     ("pad", np.uint8, 3)])


def lib():
    """Loads libefusion_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EFError(f"{LIB_PATH} not built: run `python -m elasticfusion_amd.build` (hipcc, gfx950). "
                          "There is no CPU fallback for this engine.")
        _lib = C.CDLL(LIB_PATH)
        _lib.ef_last_error.restype = C.c_char_p
        _lib.ef_last_error.argtypes = [P]
        _lib.ef_stream.restype = P
        _lib.ef_stream.argtypes = [P]
    return _lib


def use_library(path: str | None = None):
    """Rebinds this module to another build of the same C ABI (tests only: libefusion_hip_nofma.so, the -DEF_NO_FMA
    variant that is compared bit for bit with the compiled reference).  None = back to the product library."""
    global _lib, LIB_PATH
    _lib = None
    LIB_PATH = path or os.path.join(_HERE, "libefusion_hip.so")


def _chk(rc: int, ctx=None):
    if rc != 0:
        msg = lib().ef_last_error(ctx)
        raise EFError(f"libefusion_hip error {rc}: {msg.decode() if msg else ''}")


def device_count() -> int:
    n = c_i(0)
    rc = lib().ef_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def _ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(P)


class DevBuf:
    """A raw HIP allocation owned through the C ABI (ef_dev_alloc / ef_dev_free)."""

    def __init__(self, nbytes: int, fill: int | None = 0):
        self.p = P()
        self.nbytes = int(nbytes)
        _chk(lib().ef_dev_alloc(C.byref(self.p), C.c_size_t(self.nbytes)))
        if fill is not None:
            _chk(lib().ef_dev_memset(self.p, c_i(fill), C.c_size_t(self.nbytes)))

    @classmethod
    def from_array(cls, a: np.ndarray) -> "DevBuf":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes, fill=None)
        if a.nbytes:
            _chk(lib().ef_dev_upload(b.p, _ptr(a), C.c_size_t(a.nbytes)))
        return b

    def to_array(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes, (out.nbytes, self.nbytes)
        if out.nbytes:
            _chk(lib().ef_dev_download(_ptr(out), self.p, C.c_size_t(out.nbytes)))
        return out

    def __del__(self):
        if getattr(self, "p", None):
            try:
                lib().ef_dev_free(self.p)
            except Exception:
                pass
            self.p = None


def default_config(**kw) -> ef_config:
    cfg = ef_config()
    lib().ef_default_config(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise TypeError(f"unknown config field {k}")
        setattr(cfg, k, v)
    return cfg


def write_freiburg(path: str, T_wc, timestamps):
    """ef_write_freiburg on host arrays (no GPU): the reference's trajectory dump (ElasticFusion.cpp:112-139)"""
    T = np.ascontiguousarray(T_wc, np.float64).reshape(-1, 16)
    ts = np.ascontiguousarray(timestamps, np.int64)
    assert len(T) == len(ts)
    _chk(lib().ef_write_freiburg(path.encode(), _ptr(T), _ptr(ts), c_i(len(T))))


class GlobalLoop(C.Structure):   # ef_global_loop
    _fields_ = [("attempted", c_i), ("closest", c_i), ("n_constraints", c_i), ("accepted", c_i), ("graph_nodes", c_i), ("icp_error", c_f),
                ("icp_count", c_f), ("T_wc_recovery", C.c_double * 16)]


class RelocState(C.Structure):   # ef_reloc_state
    _fields_ = [("lost", c_i), ("tracking_ok", c_i), ("tracking_count", c_i), ("last_frame_recovery", c_i)]


class LocalLoop(C.Structure):   # ef_local_loop
    _fields_ = [("attempted", c_i), ("cov_ok", c_i), ("gates_ok", c_i), ("n_constraints", c_i), ("applied", c_i),
                ("graph_nodes", c_i), ("graph_capacity", c_i), ("pad_", c_i), ("stats", c_f * 6), ("cov_diag", C.c_double * 6), ("T_wc_curr", C.c_double * 16),
                ("T_wc_est", C.c_double * 16)]


LOOP_SOLVER = C.CFUNCTYPE(c_i, C.c_void_p, C.POINTER(LocalLoop), C.POINTER(C.c_double), c_i, C.POINTER(c_f), C.POINTER(c_i))


def solve_local_deformation(nodes4, constraints, src_time, last_deform_time=0):
    """ef_solve_local_deformation on host arrays (no GPU): -> (graph [n, 16] float32, error, mean constraint error) or None"""
    nodes4 = np.ascontiguousarray(nodes4, np.float32).reshape(-1, 4)
    cons = np.ascontiguousarray(constraints, np.float64).reshape(-1, 8)
    g = np.zeros((max(len(nodes4), 1), 16), np.float32)
    e, m = c_f(0), c_f(0)
    rc = lib().ef_solve_local_deformation(_ptr(nodes4), c_i(len(nodes4)), _ptr(cons), c_i(len(cons)), C.c_int64(int(src_time)),
                                          C.c_int64(int(last_deform_time)), _ptr(g), C.byref(e), C.byref(m))
    return (g[:len(nodes4)], e.value, m.value) if rc == 0 else None


class GraphConstraint(C.Structure):   # ef_graph_constraint
    _fields_ = [("src", C.c_double * 3), ("target", C.c_double * 3), ("src_time", C.c_int64), ("target_time", C.c_int64), ("relative", c_i), ("pin", c_i)]


def graph_constraints(rows):
    """rows of (src xyz, target xyz, src_time, target_time, relative, pin) -> ef_graph_constraint array"""
    arr = (GraphConstraint * max(len(rows), 1))()
    for a, (src, target, st, tt, rel, pin) in zip(arr, rows):
        a.src[:] = [float(x) for x in src]
        a.target[:] = [float(x) for x in target]
        a.src_time, a.target_time, a.relative, a.pin = int(st), int(tt), int(bool(rel)), int(bool(pin))
    return arr


def solve_deformation(nodes4, rows, fernMatch=False, last_deform_time=0, poses=None, pose_times=None, gates=None):
    """ef_solve_deformation (Deformation::constrain in general form, host only).  rows as graph_constraints takes them.
    -> dict(accepted, graph [n, 16], error, meanConsErr, poses [k, 4, 4] deformed along, new_relative rows)"""
    nodes4 = np.ascontiguousarray(nodes4, np.float32).reshape(-1, 4)
    cons = graph_constraints(rows)
    g = np.zeros((max(len(nodes4), 1), 16), np.float32)
    P16 = np.ascontiguousarray(np.zeros((0, 4, 4)) if poses is None else poses, np.float64).reshape(-1, 4, 4).copy()
    times = np.ascontiguousarray([] if pose_times is None else pose_times, np.int64)
    assert len(times) == len(P16)
    rel = (GraphConstraint * max(len(rows), 1))()
    n_rel, e, m = c_i(0), c_f(0), c_f(0)
    gz = None if gates is None else (c_f * 3)(*[float(x) for x in gates])
    rc = lib().ef_solve_deformation_gated(_ptr(nodes4), c_i(len(nodes4)), cons, c_i(len(rows)), c_i(int(bool(fernMatch))), C.c_int64(int(last_deform_time)),
                                          _ptr(P16) if len(P16) else None, _ptr(times) if len(P16) else None, c_i(len(P16)), _ptr(g), C.byref(e),
                                          C.byref(m), rel, C.byref(n_rel), gz)
    if rc not in (0, -4):
        _chk(rc)
    new_rel = [(list(r.src), list(r.target), r.src_time, r.target_time, True, False) for r in rel[:n_rel.value]]
    return dict(accepted=rc == 0, graph=g[:len(nodes4)], error=e.value, meanConsErr=m.value, poses=P16, new_relative=new_rel)


FERN_TRACKER = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(c_f), C.POINTER(c_f), C.POINTER(C.c_double), C.POINTER(c_f), C.POINTER(c_f),
                           C.POINTER(C.c_double), C.POINTER(c_f), C.POINTER(c_f))


class Ferns:
    """The reference's fern database (Core/Ferns.h:35-184) over ef_ferns_* — host-side, no GPU.  Same members where they make
    sense on arrays: addFrame, findFrame (-> T_wc_est, constraints [n, 6]), lastClosest, frames (count), conservatory (table).
    Views are the 1/8-resolution images ``ElasticFusion.imageResized(..., 8)`` returns: rgb [h, w, 3|4] uint8, verts / norms
    [h, w, 4] float32.  ``tracker(fern_verts, fern_norms, T_wc_fern, verts, norms, T_wc) -> (T_wc_est, icp_error, icp_count)``
    stands where the reference runs its 80x60 RGBDOdometry (Ferns.cpp:243-258)."""

    _prefix = "ef_ferns_"      # symbol prefix and library are overridable: a subclass can bind another implementation of the same C interface

    def _library(self):
        return lib()

    def _f(self, name):
        return getattr(self._L, self._prefix + name)

    def __init__(self, n=500, maxDepth=3000, photoThresh=115.0, width=640, height=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, seed=0):
        self._h = None
        self._L = self._library()
        f = self._f
        f("create").restype = P
        f("create").argtypes = [c_i, c_i, c_f, c_i, c_i, c_f, c_f, c_f, c_f, C.c_uint]
        f("block_hd_aware").restype = f("photometric_check").restype = c_f
        f("destroy").argtypes = [P]
        f("get_table").argtypes = f("set_table").argtypes = [P, P]
        f("add_frame").argtypes = [P, P, c_i, P, P, P, c_i, c_f]
        f("find_frame").argtypes = [P, P, c_i, P, P, P, c_i, c_i, FERN_TRACKER, P, P, P, c_i, P]
        f("count").argtypes = f("last_closest").argtypes = [P]
        f("get_frame").argtypes = [P, c_i, P, P, P, P, P, P, P]
        f("set_frame_pose").argtypes = [P, c_i, P]
        f("block_hd_aware").argtypes = [P, c_i, c_i]
        f("photometric_check").argtypes = [P, P, c_i, P, P, c_i]
        self.num, self.w, self.h = int(n), width // 8, height // 8
        self._h = f("create")(int(n), int(maxDepth), float(photoThresh), int(width), int(height), fx, fy, cx, cy, int(seed))
        if not self._h:
            raise EFError("ef_ferns_create: bad arguments")

    _owned = True

    def close(self):
        if self._h and self._owned:
            self._f("destroy")(self._h)
        self._h = None

    __del__ = close

    def _view(self, rgb, verts, norms):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        assert rgb.shape[:2] == (self.h, self.w) and rgb.shape[2] in (3, 4), rgb.shape
        verts = np.ascontiguousarray(verts, np.float32).reshape(self.h, self.w, 4)
        norms = np.ascontiguousarray(norms, np.float32).reshape(self.h, self.w, 4)
        return rgb, verts, norms

    @property
    def conservatory(self):
        t = np.zeros((self.num, 6), np.int32)
        _chk(self._f("get_table")(self._h, _ptr(t)))
        return t

    @conservatory.setter
    def conservatory(self, table):
        t = np.ascontiguousarray(table, np.int32).reshape(self.num, 6)
        _chk(self._f("set_table")(self._h, _ptr(t)))

    def addFrame(self, rgb, verts, norms, T_wc, srcTime, threshold) -> bool:
        rgb, verts, norms = self._view(rgb, verts, norms)
        T = np.ascontiguousarray(T_wc, np.float64).reshape(4, 4)
        rc = self._f("add_frame")(self._h, _ptr(rgb), rgb.shape[2], _ptr(verts), _ptr(norms), _ptr(T), int(srcTime), float(threshold))
        if rc < 0:
            _chk(rc)
        return rc == 1

    def findFrame(self, rgb, verts, norms, T_wc, time, lost, tracker):
        rgb, verts, norms = self._view(rgb, verts, norms)
        T = np.ascontiguousarray(T_wc, np.float64).reshape(4, 4)
        px = self.w * self.h * 4

        def tramp(_user, fv, fn, Tf, cv, cn, Tio, err, cnt):
            A = lambda p, n, dt: np.ctypeslib.as_array(p, shape=(n,)).astype(dt).copy()
            Te, e, k = tracker(A(fv, px, np.float32).reshape(self.h, self.w, 4), A(fn, px, np.float32).reshape(self.h, self.w, 4),
                               A(Tf, 16, np.float64).reshape(4, 4), A(cv, px, np.float32).reshape(self.h, self.w, 4),
                               A(cn, px, np.float32).reshape(self.h, self.w, 4), A(Tio, 16, np.float64).reshape(4, 4))
            Te = np.ascontiguousarray(Te, np.float64).reshape(16)
            for i in range(16):
                Tio[i] = Te[i]
            err[0], cnt[0] = float(e), float(k)

        cb = FERN_TRACKER(tramp)
        T_est = np.zeros((4, 4), np.float64)
        cons = np.zeros((self.num, 6), np.float64)
        n = c_i(0)
        rc = self._f("find_frame")(self._h, _ptr(rgb), rgb.shape[2], _ptr(verts), _ptr(norms), _ptr(T), int(time), int(bool(lost)), cb, None,
                                       _ptr(T_est), _ptr(cons), self.num, C.byref(n))
        if rc < -1:
            _chk(rc)
        return T_est, cons[:n.value].copy()

    @property
    def lastClosest(self) -> int:
        return self._f("last_closest")(self._h)

    def __len__(self):
        return self._f("count")(self._h)

    def frame(self, i):
        """-> dict(codes, goodCodes, srcTime, T_wc, rgb, verts, norms) of stored frame i"""
        codes = np.zeros(self.num, np.uint8)
        good, src = c_i(0), c_i(0)
        T = np.zeros((4, 4), np.float64)
        rgb = np.zeros((self.h, self.w, 3), np.uint8)
        verts = np.zeros((self.h, self.w, 4), np.float32)
        norms = np.zeros((self.h, self.w, 4), np.float32)
        _chk(self._f("get_frame")(self._h, int(i), _ptr(codes), C.byref(good), C.byref(src), _ptr(T), _ptr(rgb), _ptr(verts), _ptr(norms)))
        return dict(codes=codes, goodCodes=good.value, srcTime=src.value, T_wc=T, rgb=rgb, verts=verts, norms=norms)

    def setFramePose(self, i, T_wc):
        T = np.ascontiguousarray(T_wc, np.float64).reshape(4, 4)
        _chk(self._f("set_frame_pose")(self._h, int(i), _ptr(T)))

    def blockHDAware(self, a, b) -> float:
        return self._f("block_hd_aware")(self._h, int(a), int(b))

    def photometricCheck(self, rgb, verts, T_wc_est, i) -> float:
        rgb, verts, _ = self._view(rgb, verts, verts)
        T = np.ascontiguousarray(T_wc_est, np.float64).reshape(4, 4)
        return self._f("photometric_check")(self._h, _ptr(rgb), rgb.shape[2], _ptr(verts), _ptr(T), int(i))


class Closure:
    """ef_closure_*: the host side of the loop closures around the fern database (ElasticFusion.cpp:392-445, 511-526, 588-589, 609-618)."""

    def __init__(self, n=500, depthCut=3.0, photoThresh=115.0, fernThresh=0.3095, width=640, height=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, seed=0,
                 _borrowed=None):
        L = lib()
        self._owned = _borrowed is None
        L.ef_closure_create.restype = L.ef_closure_ferns.restype = P
        L.ef_closure_create.argtypes = [c_i, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_f, C.c_uint]
        L.ef_closure_destroy.argtypes = L.ef_closure_ferns.argtypes = [P]
        L.ef_closure_global.argtypes = [P, P, c_i, P, P, P, c_i, FERN_TRACKER, P, P, c_i, P, P, P]
        L.ef_closure_local.argtypes = [P, P, c_i, c_i, P, c_i, P, P]
        L.ef_closure_end_frame.argtypes = [P, P, c_i, P, P, P, c_i]
        L.ef_closure_counts.argtypes = [P, P, P, P, P]
        L.ef_closure_last_rows.argtypes = [P, P, c_i, P, P]
        L.ef_closure_relative.argtypes = [P, P, c_i]
        L.ef_closure_trajectory.argtypes = [P, P, c_i]
        self.w, self.h = width // 8, height // 8
        self._h = _borrowed if _borrowed is not None else L.ef_closure_create(int(n), depthCut, photoThresh, fernThresh, int(width), int(height), fx, fy,
                                                                              cx, cy, int(seed))
        if not self._h:
            raise EFError("ef_closure_create: bad arguments")
        self.ferns = Ferns.__new__(Ferns)        # a view of the database the closure object owns
        self.ferns._L, self.ferns._h, self.ferns._owned = L, L.ef_closure_ferns(self._h), False
        self.ferns.num, self.ferns.w, self.ferns.h = int(n), self.w, self.h
        for name, res in (("count", c_i), ("last_closest", c_i), ("block_hd_aware", c_f), ("photometric_check", c_f)):
            self.ferns._f(name).restype = res
        self.ferns._f("count").argtypes = self.ferns._f("last_closest").argtypes = [P]
        self.ferns._f("get_frame").argtypes = [P, c_i, P, P, P, P, P, P, P]
        self.ferns._f("get_table").argtypes = [P, P]

    def close(self):
        if self._h and self._owned:
            lib().ef_closure_destroy(self._h)
        self._h = None

    __del__ = close

    def _view(self, rgb, verts, norms):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        return rgb, np.ascontiguousarray(verts, np.float32).reshape(self.h, self.w, 4), np.ascontiguousarray(norms, np.float32).reshape(self.h, self.w, 4)

    def globalClosure(self, rgb, verts, norms, T_wc, tick, tracker, nodes4):
        """-> (accepted, T_recovery [4, 4], graph [nodes, 16]); tracker as in Ferns.findFrame"""
        rgb, verts, norms = self._view(rgb, verts, norms)
        T = np.ascontiguousarray(T_wc, np.float64).reshape(4, 4)
        nodes4 = np.ascontiguousarray(nodes4, np.float32).reshape(-1, 4)
        px = self.w * self.h * 4

        def tramp(_user, fv, fn, Tf, cv, cn, Tio, err, cnt):
            A = lambda p, n, dt: np.ctypeslib.as_array(p, shape=(n,)).astype(dt).copy()
            Te, e, k = tracker(A(fv, px, np.float32).reshape(self.h, self.w, 4), A(fn, px, np.float32).reshape(self.h, self.w, 4),
                               A(Tf, 16, np.float64).reshape(4, 4), A(cv, px, np.float32).reshape(self.h, self.w, 4),
                               A(cn, px, np.float32).reshape(self.h, self.w, 4), A(Tio, 16, np.float64).reshape(4, 4))
            Te = np.ascontiguousarray(Te, np.float64).reshape(16)
            for i in range(16):
                Tio[i] = Te[i]
            err[0], cnt[0] = float(e), float(k)

        cb = FERN_TRACKER(tramp)
        Tr = np.zeros((4, 4), np.float64)
        g = np.zeros((1024, 16), np.float32)
        n = c_i(0)
        rc = lib().ef_closure_global(self._h, _ptr(rgb), rgb.shape[2], _ptr(verts), _ptr(norms), _ptr(T), int(tick), cb, None, _ptr(nodes4), len(nodes4),
                                     _ptr(Tr), _ptr(g), C.byref(n))
        if rc < 0:
            _chk(rc)
        return rc == 1, Tr, g[:n.value].copy()

    def relocalise(self, rgb, verts, norms, T_wc, tick, tracker):
        """a LOST camera's mid-frame step (ef_closure_relocalise: Ferns::findFrame with lost = true) -> (found, T_recovery [4, 4])"""
        rgb, verts, norms = self._view(rgb, verts, norms)
        T = np.ascontiguousarray(T_wc, np.float64).reshape(4, 4)
        px = self.w * self.h * 4

        def tramp(_user, fv, fn, Tf, cv, cn, Tio, err, cnt):
            A = lambda p, n, dt: np.ctypeslib.as_array(p, shape=(n,)).astype(dt).copy()
            Te, e, k = tracker(A(fv, px, np.float32).reshape(self.h, self.w, 4), A(fn, px, np.float32).reshape(self.h, self.w, 4),
                               A(Tf, 16, np.float64).reshape(4, 4), A(cv, px, np.float32).reshape(self.h, self.w, 4),
                               A(cn, px, np.float32).reshape(self.h, self.w, 4), A(Tio, 16, np.float64).reshape(4, 4))
            Te = np.ascontiguousarray(Te, np.float64).reshape(16)
            for i in range(16):
                Tio[i] = Te[i]
            err[0], cnt[0] = float(e), float(k)

        cb = FERN_TRACKER(tramp)
        Tr = np.zeros((4, 4), np.float64)
        lib().ef_closure_relocalise.argtypes = [P, P, c_i, P, P, P, c_i, FERN_TRACKER, P, P]
        rc = lib().ef_closure_relocalise(self._h, _ptr(rgb), rgb.shape[2], _ptr(verts), _ptr(norms), _ptr(T), int(tick), cb, None, _ptr(Tr))
        if rc < 0:
            _chk(rc)
        return rc == 1, Tr

    def logPose(self, T_wc, tick):
        """the end of a lost camera's frame (ef_closure_log_pose): the pose joins the trajectory, no keyframe is stored"""
        T = np.ascontiguousarray(T_wc, np.float64).reshape(4, 4)
        lib().ef_closure_log_pose.argtypes = [P, P, c_i]
        _chk(lib().ef_closure_log_pose(self._h, _ptr(T), int(tick)))

    def localClosure(self, constraints8, tick, nodes4):
        cons = np.ascontiguousarray(constraints8, np.float64).reshape(-1, 8)
        nodes4 = np.ascontiguousarray(nodes4, np.float32).reshape(-1, 4)
        g = np.zeros((1024, 16), np.float32)
        n = c_i(0)
        rc = lib().ef_closure_local(self._h, _ptr(cons), len(cons), int(tick), _ptr(nodes4), len(nodes4), _ptr(g), C.byref(n))
        if rc < 0:
            _chk(rc)
        return rc == 1, g[:n.value].copy()

    def endFrame(self, rgb, verts, norms, T_wc, tick) -> bool:
        rgb, verts, norms = self._view(rgb, verts, norms)
        T = np.ascontiguousarray(T_wc, np.float64).reshape(4, 4)
        rc = lib().ef_closure_end_frame(self._h, _ptr(rgb), rgb.shape[2], _ptr(verts), _ptr(norms), _ptr(T), int(tick))
        if rc < 0:
            _chk(rc)
        return rc == 1

    def setGates(self, entry=0.06, meanConsErr=3e-4, energy=0.12):
        """the three gates of the global deformation (ef_closure_set_gates); defaults = the reference's constants"""
        _chk(lib().ef_closure_set_gates(P(self._h) if not isinstance(self._h, P) else self._h, c_f(entry), c_f(meanConsErr), c_f(energy)))

    def counts(self):
        v = [c_i(0) for _ in range(4)]
        _chk(lib().ef_closure_counts(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("deforms", "fernDeforms", "relative", "trajectory"), (x.value for x in v)))

    @staticmethod
    def _rows(arr, n):
        return np.array([list(r.src) + list(r.target) + [r.src_time, r.target_time, r.relative, r.pin] for r in arr[:n]], np.float64).reshape(-1, 10)

    def lastRows(self):
        """-> (rows [n, 10] handed to the optimiser by the last closure, its energy, its mean constraint error)"""
        n = lib().ef_closure_last_rows(self._h, None, 0, None, None)
        arr = (GraphConstraint * max(n, 1))()
        e, m = c_f(0), c_f(0)
        lib().ef_closure_last_rows(self._h, arr, n, C.byref(e), C.byref(m))
        return self._rows(arr, n), e.value, m.value

    def relativeConstraints(self):
        n = lib().ef_closure_relative(self._h, None, 0)
        arr = (GraphConstraint * max(n, 1))()
        lib().ef_closure_relative(self._h, arr, n)
        return self._rows(arr, n)

    def trajectory(self):
        n = lib().ef_closure_trajectory(self._h, None, 0)
        T = np.zeros((max(n, 1), 4, 4), np.float64)
        lib().ef_closure_trajectory(self._h, _ptr(T), n)
        return T[:n]


class ElasticFusion:
    """Mirror of ``class ElasticFusion`` (Core/ElasticFusion.h) over the C ABI."""

    IMAGES = dict(depth_filtered=(0, np.uint16, 1), depth_metric=(1, np.float32, 1), depth_metric_filtered=(2, np.float32, 1),
                  image=(3, np.uint8, 4), vertex=(4, np.float32, 4), normal=(5, np.float32, 4), time=(6, np.uint16, 1),
                  fill_image=(7, np.uint8, 4), fill_vertex=(8, np.float32, 4), fill_normal=(9, np.float32, 4),
                  index=(10, np.uint32, 1), vertConf=(11, np.float32, 4), colorTime=(12, np.float32, 4), normRad=(13, np.float32, 4),
                  old_image=(14, np.uint8, 4), old_vertex=(15, np.float32, 4), old_normal=(16, np.float32, 4), old_time=(17, np.uint16, 1))
    TRACKER = dict(vmap_curr=(0, np.float32, 3), nmap_curr=(1, np.float32, 3), vmap_g_prev=(2, np.float32, 3),
                   nmap_g_prev=(3, np.float32, 3), lastDepth=(4, np.float32, 1), nextDepth=(5, np.float32, 1),
                   lastImage=(6, np.uint8, 1), nextImage=(7, np.uint8, 1), lastNextImage=(8, np.uint8, 1),
                   dIdx=(9, np.int16, 1), dIdy=(10, np.int16, 1), depth_tmp=(11, np.uint16, 1))

    def __init__(self, width=640, height=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, timeDelta=2147483647 // 2,
                 confidence=10.0, depthCut=3.0, icpThresh=10.0, fastOdom=False, so3=True, frameToFrameRGB=False,
                 closeLoops=False, maxSurfels=4 * 1024 * 1024, device=0, stream=None, countThresh=35000, errThresh=5e-05,
                 covThresh=1e-05, reloc=False):
        cfg = default_config(width=width, height=height, fx=fx, fy=fy, cx=cx, cy=cy, time_delta=timeDelta,
                             confidence=confidence, depth_cut=depthCut, icp_weight=icpThresh, fast_odom=int(fastOdom),
                             so3=int(so3), frame_to_frame_rgb=int(frameToFrameRGB), close_loops=int(closeLoops),
                             max_surfels=maxSurfels, device=device, stream=stream)
        self.cfg = cfg
        self.h = P()
        _chk(lib().ef_create(C.byref(cfg), C.byref(self.h)))
        self._solver = None
        if closeLoops:
            _chk(lib().ef_set_loop_thresholds(self.h, c_i(countThresh), c_f(errThresh), c_f(covThresh)), self.h)
        if reloc:
            self.setRelocalisation(True)

    def close(self):
        if getattr(self, "h", None):
            lib().ef_destroy(self.h)
            self.h = None

    __del__ = close

    # --- frame tier ---
    def processFrame(self, rgb: np.ndarray, depth: np.ndarray, timestamp: int = 0, weightMultiplier: float = 1.0, in_T_wc=None):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        assert rgb.size == self.cfg.width * self.cfg.height * 3 and depth.size == self.cfg.width * self.cfg.height
        T = None if in_T_wc is None else np.ascontiguousarray(in_T_wc, np.float64).reshape(16)
        _chk(lib().ef_process_frame(self.h, _ptr(rgb), _ptr(depth), C.c_int64(timestamp), c_f(weightMultiplier),
                                    _ptr(T) if T is not None else None), self.h)

    def processFrameDevice(self, rgb_dev, depth_dev, timestamp: int = 0, weightMultiplier: float = 1.0, in_T_wc=None):
        """rgb_dev / depth_dev: raw device pointers (int or c_void_p) of frames already resident in HBM."""
        T = None if in_T_wc is None else np.ascontiguousarray(in_T_wc, np.float64).reshape(16)
        _chk(lib().ef_process_frame_dev(self.h, P(int(rgb_dev)), P(int(depth_dev)), C.c_int64(timestamp), c_f(weightMultiplier),
                                        _ptr(T) if T is not None else None), self.h)

    # --- local loop closure, front half (ElasticFusion.cpp:447-527; closeLoops=True contexts) ---
    def setLoopSolver(self, fn):
        """fn(info: LocalLoop, constraints [n, 8] float64) -> None (reject) | graph [nodes, 16] float32 (accept).
        Stands where Deformation::constrain stands in the reference; called inside processFrame."""
        if fn is None:
            self._solver = None
            _chk(lib().ef_set_loop_solver(self.h, None, None), self.h)
            return

        def tramp(user, info, cons, n, graph_out, nodes_out):
            c = np.ctypeslib.as_array(cons, shape=(n, 8)).copy() if n > 0 else np.zeros((0, 8))
            g = fn(info.contents, c)
            if g is None:
                return 0
            g = np.ascontiguousarray(g, np.float32).reshape(-1, 16)
            nodes_out[0] = len(g)
            if len(g) > info.contents.graph_capacity:   # graph_out only has room for graph_capacity nodes: let the engine reject the count
                return 1
            C.memmove(graph_out, g.ctypes.data, g.nbytes)
            return 1
        self._solver = LOOP_SOLVER(tramp)
        _chk(lib().ef_set_loop_solver(self.h, self._solver, None), self.h)

    # --- relocalisation (the reference constructor's `reloc`; ElasticFusion.cpp:326-366, 411-413, 536, 601-604) ---
    def setRelocalisation(self, on=True):
        _chk(lib().ef_set_relocalisation(self.h, c_i(int(on))), self.h)

    def relocState(self) -> RelocState:
        s = RelocState()
        _chk(lib().ef_get_relocalisation(self.h, C.byref(s)), self.h)
        return s

    def getLost(self) -> bool:
        return bool(self.relocState().lost)

    # --- global loop closure (ElasticFusion.cpp:392-445, 609-618; closeLoops=True contexts) ---
    def enableGlobalClosure(self, n=500, photoThresh=115.0, fernThresh=0.3095, seed=0):
        """ef_enable_global_closure: the fern database, its 1/8-resolution tracker and the closure bookkeeping inside processFrame"""
        _chk(lib().ef_enable_global_closure(self.h, c_i(int(n)), c_f(photoThresh), c_f(fernThresh), C.c_uint(int(seed))), self.h)
        lib().ef_get_closure.restype = P
        lib().ef_get_closure.argtypes = [P]
        self._closure = Closure(n=n, width=self.cfg.width, height=self.cfg.height, _borrowed=lib().ef_get_closure(self.h))
        return self._closure

    def getFerns(self):
        """the fern database of the context (Ferns view: len() = frames.size(), lastClosest(), frame(i))"""
        return self.closure().ferns

    def closure(self):
        # every look at the closure object goes through ef_get_closure: the engine defers the end-of-frame bookkeeping (keyframe decision,
        # trajectory entry) to its next synchronisation point, and ef_get_closure is one
        lib().ef_get_closure.restype = P
        lib().ef_get_closure.argtypes = [P]
        lib().ef_get_closure(self.h)
        return self._closure

    def globalLoop(self) -> GlobalLoop:
        g = GlobalLoop()
        _chk(lib().ef_get_global_loop(self.h, C.byref(g)), self.h)
        return g

    def useBuiltinLoopSolver(self, on=True):
        """the built-in deformation-graph optimiser where Deformation::constrain stands (ef_use_builtin_loop_solver)"""
        _chk(lib().ef_use_builtin_loop_solver(self.h, c_i(int(on))), self.h)

    def localLoop(self):
        """(info, constraints [n, 8]) of the last processFrame."""
        info = LocalLoop()
        cons = np.zeros((4096, 8), np.float64)
        n = c_i(0)
        _chk(lib().ef_get_local_loop(self.h, C.byref(info), _ptr(cons), c_i(len(cons)), C.byref(n)), self.h)
        return info, cons[:n.value].copy()

    def imageResized(self, name: str, factor: int) -> np.ndarray:
        """Resize::{image,vertex,time}: the named predicted / fill-in / inactive image, NEAREST-downsampled on the device."""
        which, dt, ch = self.IMAGES[name]
        h, w = self.cfg.height // factor, self.cfg.width // factor
        out = np.zeros((h, w, ch) if ch > 1 else (h, w), dt)
        _chk(lib().ef_get_image_resized(self.h, c_i(which), c_i(factor), _ptr(out), C.c_size_t(out.nbytes)), self.h)
        return out

    def sampleGraph(self, max_nodes=1024):
        """Deformation::sampleGraphModel: [n, 4] float32 {x, y, z, initTime} of every 5000th surfel."""
        out = np.zeros((max_nodes, 4), np.float32)
        n = c_i(0)
        _chk(lib().ef_sample_graph(self.h, _ptr(out), c_i(max_nodes), C.byref(n)), self.h)
        return out[:n.value].copy()

    def predict(self):
        _chk(lib().ef_predict(self.h), self.h)

    def setResidentLevels(self, on=True):
        """level-resident pixel data in the persistent tracker launch (ef_set_resident_levels; default on)"""
        _chk(lib().ef_set_resident_levels(self.h, c_i(int(on))), self.h)

    def setFusedStep(self, on=True):
        """level-0 update step inside the correspondence-search launch (ef_set_fused_step)"""
        _chk(lib().ef_set_fused_step(self.h, c_i(int(on))), self.h)

    def setTrackOnly(self, on=True):
        """odometry on a frozen map: track + predict, no fusion (ef_set_track_only; BASELINE.json configs[4])"""
        _chk(lib().ef_set_track_only(self.h, c_i(int(on))), self.h)

    def setPersistentTracker(self, on=True):
        """small pyramid levels + SO(3) in one persistent launch (default) or one launch per step (ef_set_persistent_tracker)"""
        _chk(lib().ef_set_persistent_tracker(self.h, c_i(int(on))), self.h)

    def debugInjectTrackerAbort(self):
        """test hook (ef_debug_inject_tracker_abort): what a persistent tracker launch leaves when a wait timed out after admission"""
        _chk(lib().ef_debug_inject_tracker_abort(self.h), self.h)

    def synchronize(self):
        _chk(lib().ef_synchronize(self.h), self.h)

    def trackerFallbacks(self) -> int:
        """persistent tracker launches that ran on one workgroup because other work held part of the chip (ef_get_tracker_fallbacks)"""
        n = c_i(0)
        _chk(lib().ef_get_tracker_fallbacks(self.h, C.byref(n)), self.h)
        return n.value

    def debugOccupy(self, workgroups: int, microseconds: int):
        """test hook: CU-filling workgroups spinning on a stream of their own (ef_debug_occupy)"""
        _chk(lib().ef_debug_occupy(self.h, c_i(int(workgroups)), c_i(int(microseconds))), self.h)

    def stream(self) -> int:
        return int(lib().ef_stream(self.h) or 0)

    def get_T_wc(self) -> np.ndarray:
        T = np.zeros(16, np.float64)
        _chk(lib().ef_get_pose(self.h, _ptr(T)), self.h)
        return T.reshape(4, 4)

    def getTick(self) -> int:
        t = c_i(0)
        _chk(lib().ef_get_tick(self.h, C.byref(t)), self.h)
        return t.value

    def setTick(self, v: int):
        _chk(lib().ef_set_tick(self.h, c_i(v)), self.h)

    def getTimeDelta(self) -> int:
        return self.cfg.time_delta

    def getConfidenceThreshold(self) -> float:
        return self.cfg.confidence

    def getMaxDepthProcessed(self) -> float:
        return 20.0

    def trackingStats(self):
        out = np.zeros(6, np.float32)
        A = np.zeros((6, 6), np.float64)
        b = np.zeros(6, np.float64)
        _chk(lib().ef_get_tracking_stats(self.h, _ptr(out), _ptr(A), _ptr(b)), self.h)
        return out, A, b

    def getCovariance(self) -> np.ndarray:
        c = np.zeros(36, np.float64)
        _chk(lib().ef_get_covariance(self.h, _ptr(c)), self.h)
        return c.reshape(6, 6)

    def trajectory(self):
        n = c_i(0)
        _chk(lib().ef_get_trajectory(self.h, None, None, c_i(2**31 - 1), C.byref(n)), self.h)   # how many poses are logged (one per processed frame)
        cap = max(int(n.value), 1)
        T = np.zeros((cap, 16), np.float64)
        ts = np.zeros(cap, np.int64)
        _chk(lib().ef_get_trajectory(self.h, _ptr(T), _ptr(ts), c_i(cap), C.byref(n)), self.h)
        return T[:n.value].reshape(-1, 4, 4).copy(), ts[:n.value].copy()

    def lastCount(self) -> int:
        n = c_u32(0)
        _chk(lib().ef_map_count(self.h, C.byref(n)), self.h)
        return n.value

    def downloadMap(self) -> np.ndarray:
        n = self.lastCount()
        out = np.zeros((max(n, 1), 12), np.float32)
        got = c_u32(0)
        _chk(lib().ef_map_download(self.h, _ptr(out), c_u32(n), C.byref(got)), self.h)
        return out[:got.value].copy()

    def setReferenceDownload(self, on=True):
        """downloadMap / savePly read what GlobalModel::downloadMap reads (the pre-clean buffer, quirk Q14) instead of model()"""
        _chk(lib().ef_set_reference_download(self.h, c_i(int(on))), self.h)

    def uploadMap(self, surfels: np.ndarray):
        s = np.ascontiguousarray(surfels, np.float32).reshape(-1, 12)
        _chk(lib().ef_map_upload(self.h, _ptr(s), c_u32(len(s))), self.h)

    def getPoseQT(self) -> np.ndarray:
        """T_wc as the engine holds it: unit quaternion x y z w + translation (7 doubles), no matrix round trip (ef_get_pose_qt)"""
        qt = np.zeros(7, np.float64)
        _chk(lib().ef_get_pose_qt(self.h, _ptr(qt)), self.h)
        return qt

    def checkpoint(self, last_rgb, last_depth) -> dict:
        """what a replay carries from one processFrame to the next: map, tick, pose and the frame processed last"""
        return dict(map=self.downloadMap(), tick=self.getTick(), qt=self.getPoseQT(), rgb=np.ascontiguousarray(last_rgb, np.uint8).copy(),
                    depth=np.ascontiguousarray(last_depth, np.uint16).copy())

    def restore(self, ck: dict):
        """resume from ``checkpoint()`` in a fresh context (ef_map_upload + ef_restore_state): the next processFrame continues the replay"""
        self.uploadMap(ck["map"])
        qt = np.ascontiguousarray(ck["qt"], np.float64).reshape(7)
        _chk(lib().ef_restore_state(self.h, c_i(int(ck["tick"])), _ptr(qt), _ptr(ck["rgb"]), _ptr(ck["depth"])), self.h)

    def savePly(self, path: str):
        _chk(lib().ef_save_ply(self.h, path.encode()), self.h)

    def saveFreiburg(self, path: str):
        _chk(lib().ef_save_freiburg(self.h, path.encode()), self.h)

    def setRgbOnly(self, v): _chk(lib().ef_set_rgb_only(self.h, c_i(int(v))), self.h)
    def setIcpWeight(self, v): _chk(lib().ef_set_icp_weight(self.h, c_f(v)), self.h)
    def setPyramid(self, v): _chk(lib().ef_set_pyramid(self.h, c_i(int(v))), self.h)
    def setFastOdom(self, v): _chk(lib().ef_set_fast_odom(self.h, c_i(int(v))), self.h)
    def setSo3(self, v): _chk(lib().ef_set_so3(self.h, c_i(int(v))), self.h)
    def setFrameToFrameRGB(self, v): _chk(lib().ef_set_frame_to_frame_rgb(self.h, c_i(int(v))), self.h)
    def setConfidenceThreshold(self, v): _chk(lib().ef_set_confidence_threshold(self.h, c_f(v)), self.h)
    def setDepthCutoff(self, v): _chk(lib().ef_set_depth_cutoff(self.h, c_f(v)), self.h)
    def setInputOverlap(self, on): _chk(lib().ef_set_input_overlap(self.h, c_i(int(on))), self.h)
    def setInputCuMask(self, one_in_n): _chk(lib().ef_set_input_cu_mask(self.h, c_i(int(one_in_n))), self.h)
    def setDeformation(self, graph, isFern=False):
        g = np.ascontiguousarray(graph, np.float32).reshape(-1, 16)
        _chk(lib().ef_set_deformation(self.h, _ptr(g), c_i(len(g)), c_i(int(isFern))), self.h)

    def setGraphReplay(self, on): _chk(lib().ef_set_graph_replay(self.h, c_i(int(on))), self.h)

    def image(self, name: str) -> np.ndarray:
        which, dt, ch = self.IMAGES[name]
        W, H = self.cfg.width, self.cfg.height
        out = np.zeros((H, W, ch) if ch > 1 else (H, W), dt)
        _chk(lib().ef_get_image(self.h, c_i(which), _ptr(out), C.c_size_t(out.nbytes)), self.h)
        return out

    def trackerBuffer(self, name: str, level: int = 0) -> np.ndarray:
        which, dt, planes = self.TRACKER[name]
        w, h = self.cfg.width >> level, self.cfg.height >> level
        out = np.zeros((h * planes, w), dt)
        _chk(lib().ef_get_tracker_buffer(self.h, c_i(which), c_i(level), _ptr(out), C.c_size_t(out.nbytes)), self.h)
        return out

    def enableTiming(self, on=True):
        _chk(lib().ef_enable_timing(self.h, c_i(int(on))), self.h)

    def timings(self) -> dict:
        arr = (ef_timing * 32)()
        n = c_i(0)
        _chk(lib().ef_get_timings(self.h, arr, c_i(32), C.byref(n)), self.h)
        return {arr[i].name.decode(): arr[i].ms for i in range(n.value)}


# ------------------------------------------------------------------------------------------------
# operator tier on numpy arrays (upload -> HIP kernel via the C ABI -> download)
# ------------------------------------------------------------------------------------------------
class ops:
    @staticmethod
    def _f32(a):
        return np.ascontiguousarray(a, np.float32)

    @staticmethod
    def pyr_down(src):
        h, w = src.shape
        s, d = DevBuf.from_array(src), DevBuf((h // 2) * (w // 2) * 2)
        _chk(lib().ef_op_pyr_down(s.p, c_i(w), c_i(h), d.p, None))
        return d.to_array(np.uint16, (h // 2, w // 2))

    @staticmethod
    def create_vmap(depth, fx, fy, cx, cy, cutoff, init=None):
        h, w = depth.shape
        d = DevBuf.from_array(depth)
        v = DevBuf.from_array(np.zeros((3 * h, w), np.float32) if init is None else init)
        k = ef_intr(fx, fy, cx, cy)
        _chk(lib().ef_op_create_vmap(C.byref(k), d.p, c_i(w), c_i(h), c_f(cutoff), v.p, None))
        return v.to_array(np.float32, (3 * h, w))

    @staticmethod
    def create_nmap(vmap, init=None):
        h3, w = vmap.shape
        v = DevBuf.from_array(vmap)
        n = DevBuf.from_array(np.zeros((h3, w), np.float32) if init is None else init)
        _chk(lib().ef_op_create_nmap(v.p, c_i(w), c_i(h3 // 3), n.p, None))
        return n.to_array(np.float32, (h3, w))

    @staticmethod
    def transform_maps(vmap, nmap, R, t):
        h3, w = vmap.shape
        v, n = DevBuf.from_array(vmap), DevBuf.from_array(nmap)
        R9, t3 = ops._f32(R).reshape(9), ops._f32(t).reshape(3)
        _chk(lib().ef_op_transform_maps(v.p, n.p, c_i(w), c_i(h3 // 3), _ptr(R9), _ptr(t3), v.p, n.p, None))
        return v.to_array(np.float32, (h3, w)), n.to_array(np.float32, (h3, w))

    @staticmethod
    def copy_maps(vtex, ntex):
        h, w, _ = vtex.shape
        v4, n4 = DevBuf.from_array(ops._f32(vtex)), DevBuf.from_array(ops._f32(ntex))
        tmp, vm, nm = DevBuf(h * w * 16), DevBuf(h * w * 12), DevBuf(h * w * 12)
        _chk(lib().ef_op_copy_maps(v4.p, n4.p, c_i(w), c_i(h), tmp.p, vm.p, nm.p, None))
        return tmp.to_array(np.float32, (h, w, 4)), vm.to_array(np.float32, (3 * h, w)), nm.to_array(np.float32, (3 * h, w))

    @staticmethod
    def resize_map(src, normalize, init=None):
        h3, w = src.shape
        s = DevBuf.from_array(src)
        o = DevBuf.from_array(np.zeros((h3 // 2, w // 2), np.float32) if init is None else init)
        fn = lib().ef_op_resize_nmap if normalize else lib().ef_op_resize_vmap
        _chk(fn(s.p, c_i(w), c_i(h3 // 3), o.p, None))
        return o.to_array(np.float32, (h3 // 2, w // 2))

    @staticmethod
    def pyr_down_gauss_f(src):
        h, w = src.shape
        s, d = DevBuf.from_array(src), DevBuf((h // 2) * (w // 2) * 4)
        _chk(lib().ef_op_pyr_down_gauss_f(s.p, c_i(w), c_i(h), d.p, None))
        return d.to_array(np.float32, (h // 2, w // 2))

    @staticmethod
    def pyr_down_uchar_gauss(src):
        h, w = src.shape
        s, d = DevBuf.from_array(src), DevBuf((h // 2) * (w // 2))
        _chk(lib().ef_op_pyr_down_uchar_gauss(s.p, c_i(w), c_i(h), d.p, None))
        return d.to_array(np.uint8, (h // 2, w // 2))

    @staticmethod
    def vertices_to_depth(vmaps_tmp, cutoff):
        h, w, _ = vmaps_tmp.shape
        s, d = DevBuf.from_array(ops._f32(vmaps_tmp)), DevBuf(h * w * 4)
        _chk(lib().ef_op_vertices_to_depth(s.p, c_i(w), c_i(h), c_f(cutoff), d.p, None))
        return d.to_array(np.float32, (h, w))

    @staticmethod
    def bgr_to_intensity(rgba):
        h, w, _ = rgba.shape
        s, d = DevBuf.from_array(rgba), DevBuf(h * w)
        _chk(lib().ef_op_image_bgr_to_intensity(s.p, c_i(w), c_i(h), d.p, None))
        return d.to_array(np.uint8, (h, w))

    @staticmethod
    def derivative_images(img):
        h, w = img.shape
        s, dx, dy = DevBuf.from_array(img), DevBuf(h * w * 2), DevBuf(h * w * 2)
        _chk(lib().ef_op_compute_derivative_images(s.p, c_i(w), c_i(h), dx.p, dy.p, None))
        return dx.to_array(np.int16, (h, w)), dy.to_array(np.int16, (h, w))

    @staticmethod
    def project_to_point_cloud(depth, fx, fy, cx, cy, level=0):
        h, w = depth.shape
        s, d = DevBuf.from_array(depth), DevBuf(h * w * 12)
        k = ef_intr(fx, fy, cx, cy)
        _chk(lib().ef_op_project_to_point_cloud(s.p, c_i(w), c_i(h), C.byref(k), c_i(level), d.p, None))
        return d.to_array(np.float32, (h, w, 3))

    @staticmethod
    def icp_step(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, distThres, angleThres):
        h3, w = vmap_curr.shape
        bufs = [DevBuf.from_array(a) for a in (vmap_curr, nmap_curr, vmap_g_prev, nmap_g_prev)]
        A, b, res = np.zeros((6, 6), np.float32), np.zeros(6, np.float32), np.zeros(2, np.float32)
        k = ef_intr(*intr)
        _chk(lib().ef_op_icp_step(_ptr(ops._f32(Rcurr).reshape(9)), _ptr(ops._f32(tcurr)), bufs[0].p, bufs[1].p,
                                  _ptr(ops._f32(Rprev_inv).reshape(9)), _ptr(ops._f32(tprev)), C.byref(k), bufs[2].p, bufs[3].p,
                                  c_f(distThres), c_f(angleThres), c_i(w), c_i(h3 // 3), _ptr(A), _ptr(b), _ptr(res), None))
        return A, b, res

    @staticmethod
    def rgb_residual(minScale, dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage, maxDepthDelta, kt, krkinv):
        h, w = nextImage.shape
        bufs = [DevBuf.from_array(a) for a in (dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage)]
        corres = DevBuf(h * w * 16)
        sigma, count = c_i(0), c_i(0)
        _chk(lib().ef_op_compute_rgb_residual(c_f(minScale), bufs[0].p, bufs[1].p, bufs[2].p, bufs[3].p, bufs[4].p, bufs[5].p, corres.p,
                                              c_f(maxDepthDelta), _ptr(ops._f32(kt)), _ptr(ops._f32(krkinv).reshape(9)), c_i(w), c_i(h),
                                              C.byref(sigma), C.byref(count), None))
        return corres.to_array(DATATERM, (h, w)), sigma.value, count.value

    @staticmethod
    def rgb_step(corres, sigma, cloud, fx, fy, dIdx, dIdy, sobelScale):
        h, w = corres.shape
        bufs = [DevBuf.from_array(a) for a in (corres, ops._f32(cloud), dIdx, dIdy)]
        A, b = np.zeros((6, 6), np.float32), np.zeros(6, np.float32)
        _chk(lib().ef_op_rgb_step(bufs[0].p, c_f(sigma), bufs[1].p, c_f(fx), c_f(fy), bufs[2].p, bufs[3].p, c_f(sobelScale), c_i(w), c_i(h),
                                  _ptr(A), _ptr(b), None))
        return A, b

    @staticmethod
    def so3_step(lastImage, nextImage, imageBasis, kinv, krlr):
        h, w = nextImage.shape
        li, ni = DevBuf.from_array(lastImage), DevBuf.from_array(nextImage)
        A, b, res = np.zeros((3, 3), np.float32), np.zeros(3, np.float32), np.zeros(2, np.float32)
        _chk(lib().ef_op_so3_step(li.p, ni.p, _ptr(ops._f32(imageBasis).reshape(9)), _ptr(ops._f32(kinv).reshape(9)),
                                  _ptr(ops._f32(krlr).reshape(9)), c_i(w), c_i(h), _ptr(A), _ptr(b), _ptr(res), None))
        return A, b, res

    @staticmethod
    def clean_deform(cam, T_wc, time, idx, vc, ct, nr, confThreshold, timeDelta, maxDepth, surfels, newUnstable, graph, depth, isFern=0):
        s = ops._f32(surfels).reshape(-1, 12)
        nu = ops._f32(newUnstable).reshape(-1, 12)
        g = ops._f32(graph).reshape(-1, 16)
        bufs = [DevBuf.from_array(a) for a in (idx, ops._f32(vc), ops._f32(ct), ops._f32(nr), g, ops._f32(depth))]
        sb, nb = DevBuf.from_array(s), DevBuf.from_array(nu if len(nu) else np.zeros((1, 12), np.float32))
        out = DevBuf((len(s) + len(nu) + 1) * 48)
        n = c_u32(0)
        _chk(lib().ef_op_clean_deform(C.byref(cam), _ptr(ops._T(T_wc)), c_i(time), bufs[0].p, bufs[1].p, bufs[2].p, bufs[3].p,
                                      c_f(confThreshold), c_i(timeDelta), c_f(maxDepth), sb.p, c_u32(len(s)), nb.p, c_u32(len(nu)),
                                      bufs[4].p, c_i(len(g)), bufs[5].p, c_i(int(isFern)), out.p, C.byref(n), None))
        return out.to_array(np.float32, (n.value, 12))

    LINALG = dict(ldlt6=0, ldlt3f=1, polar3=2, rodrigues=3, se3_inverse=4, se3_log_norm=5, scalar=6, ldlt6_wave=7)

    @staticmethod
    def linalg(which, vec, n_out):
        """The driver's Eigen/Sophus arithmetic as the device evaluates it (ef_op_linalg)."""
        v = np.ascontiguousarray(np.asarray(vec, np.float64).reshape(-1))
        out = np.zeros(n_out, np.float64)
        _chk(lib().ef_op_linalg(c_i(ops.LINALG[which]), _ptr(v), c_i(v.size), _ptr(out), c_i(n_out)))
        return out

    @staticmethod
    def filter_depth(raw, maxD):
        h, w = raw.shape
        s, d = DevBuf.from_array(raw), DevBuf(h * w * 2)
        _chk(lib().ef_op_filter_depth(s.p, c_i(w), c_i(h), c_f(maxD), d.p, None))
        return d.to_array(np.uint16, (h, w))

    @staticmethod
    def metricise_depth(d_in, maxD):
        h, w = d_in.shape
        s, d = DevBuf.from_array(d_in), DevBuf(h * w * 4)
        _chk(lib().ef_op_metricise_depth(s.p, c_i(w), c_i(h), c_f(maxD), d.p, None))
        return d.to_array(np.float32, (h, w))

    @staticmethod
    def seed_map(cam, rgb, dm, dmf, time, maxDepth):
        Pn = cam.cols * cam.rows
        bufs = [DevBuf.from_array(a) for a in (rgb, dm, dmf)]
        out = DevBuf(Pn * 48)
        n = c_u32(0)
        _chk(lib().ef_op_seed_map(C.byref(cam), bufs[0].p, bufs[1].p, bufs[2].p, c_i(time), c_f(maxDepth), out.p, C.byref(n), None))
        return out.to_array(np.float32, (n.value, 12))

    @staticmethod
    def _T(T):
        return np.ascontiguousarray(T, np.float64).reshape(16)

    @staticmethod
    def predict_indices(cam, T_wc, time, surfels, maxDepth, timeDelta):
        s = ops._f32(surfels).reshape(-1, 12)
        Pn = cam.cols * cam.rows
        sb = DevBuf.from_array(s)
        idx, vc, ct, nr = DevBuf(Pn * 4), DevBuf(Pn * 16), DevBuf(Pn * 16), DevBuf(Pn * 16)
        _chk(lib().ef_op_predict_indices(C.byref(cam), _ptr(ops._T(T_wc)), c_i(time), sb.p, c_u32(len(s)), c_f(maxDepth), c_i(timeDelta),
                                         idx.p, vc.p, ct.p, nr.p, None))
        shp = (cam.rows, cam.cols)
        return (idx.to_array(np.uint32, shp), vc.to_array(np.float32, shp + (4,)), ct.to_array(np.float32, shp + (4,)),
                nr.to_array(np.float32, shp + (4,)))

    @staticmethod
    def combined_predict(cam, T_wc, surfels, maxDepth, confThreshold, time, maxTime, timeDelta):
        s = ops._f32(surfels).reshape(-1, 12)
        Pn = cam.cols * cam.rows
        sb = DevBuf.from_array(s)
        img, vt, nm, tm = DevBuf(Pn * 4), DevBuf(Pn * 16), DevBuf(Pn * 16), DevBuf(Pn * 2)
        _chk(lib().ef_op_combined_predict(C.byref(cam), _ptr(ops._T(T_wc)), sb.p, c_u32(len(s)), c_f(maxDepth), c_f(confThreshold),
                                          c_i(time), c_i(maxTime), c_i(timeDelta), img.p, vt.p, nm.p, tm.p, None))
        shp = (cam.rows, cam.cols)
        return (img.to_array(np.uint8, shp + (4,)), vt.to_array(np.float32, shp + (4,)), nm.to_array(np.float32, shp + (4,)),
                tm.to_array(np.uint16, shp))

    @staticmethod
    def synthesize_depth(cam, T_wc, surfels, maxDepth, confThreshold, time, maxTime, timeDelta):
        s = ops._f32(surfels).reshape(-1, 12)
        Pn = cam.cols * cam.rows
        sb, d = DevBuf.from_array(s), DevBuf(Pn * 4)
        _chk(lib().ef_op_synthesize_depth(C.byref(cam), _ptr(ops._T(T_wc)), sb.p, c_u32(len(s)), c_f(maxDepth), c_f(confThreshold),
                                          c_i(time), c_i(maxTime), c_i(timeDelta), d.p, None))
        return d.to_array(np.float32, (cam.rows, cam.cols))

    @staticmethod
    def fill_in(cam, image, vertex, normal, depthFiltered, rgb, passthrough=0, passthroughImage=0):
        bufs = [DevBuf.from_array(a) for a in (image, ops._f32(vertex), ops._f32(normal), depthFiltered, rgb)]
        Pn = cam.cols * cam.rows
        fi, fv, fn = DevBuf(Pn * 4), DevBuf(Pn * 16), DevBuf(Pn * 16)
        _chk(lib().ef_op_fill_in(C.byref(cam), bufs[0].p, bufs[1].p, bufs[2].p, bufs[3].p, bufs[4].p, c_i(passthrough), c_i(passthroughImage),
                                 fi.p, fv.p, fn.p, None))
        shp = (cam.rows, cam.cols)
        return fi.to_array(np.uint8, shp + (4,)), fv.to_array(np.float32, shp + (4,)), fn.to_array(np.float32, shp + (4,))

    @staticmethod
    def dense_enough(cam, image):
        b = DevBuf.from_array(image)
        d = c_i(0)
        _chk(lib().ef_op_dense_enough(C.byref(cam), b.p, C.byref(d), None))
        return bool(d.value)

    @staticmethod
    def fuse(cam, T_wc, time, rgb, dm, dmf, idx, vc, ct, nr, maxDepth, weighting, surfels):
        s = ops._f32(surfels).reshape(-1, 12)
        bufs = [DevBuf.from_array(a) for a in (rgb, dm, dmf, idx, ops._f32(vc), ops._f32(ct), ops._f32(nr))]
        sb = DevBuf.from_array(s)
        nu = DevBuf((cam.cols // 2) * (cam.rows // 2) * 48)
        n = c_u32(0)
        _chk(lib().ef_op_fuse(C.byref(cam), _ptr(ops._T(T_wc)), c_i(time), bufs[0].p, bufs[1].p, bufs[2].p, bufs[3].p, bufs[4].p, bufs[5].p,
                              bufs[6].p, c_f(maxDepth), c_f(weighting), sb.p, c_u32(len(s)), nu.p, C.byref(n), None))
        return sb.to_array(np.float32, (len(s), 12)), nu.to_array(np.float32, (n.value, 12))

    @staticmethod
    def clean(cam, T_wc, time, idx, vc, ct, nr, confThreshold, timeDelta, maxDepth, surfels, newUnstable):
        s = ops._f32(surfels).reshape(-1, 12)
        nu = ops._f32(newUnstable).reshape(-1, 12)
        bufs = [DevBuf.from_array(a) for a in (idx, ops._f32(vc), ops._f32(ct), ops._f32(nr))]
        sb, nb = DevBuf.from_array(s), DevBuf.from_array(nu if len(nu) else np.zeros((1, 12), np.float32))
        out = DevBuf((len(s) + len(nu) + 1) * 48)
        n = c_u32(0)
        _chk(lib().ef_op_clean(C.byref(cam), _ptr(ops._T(T_wc)), c_i(time), bufs[0].p, bufs[1].p, bufs[2].p, bufs[3].p, c_f(confThreshold),
                               c_i(timeDelta), c_f(maxDepth), sb.p, c_u32(len(s)), nb.p, c_u32(len(nu)), out.p, C.byref(n), None))
        return out.to_array(np.float32, (n.value, 12))
