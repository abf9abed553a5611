"""ctypes binding of the CPU oracle (oracle/libefo_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

c_f = C.c_float
c_i = C.c_int
P = C.c_void_p


def build(force: bool = False) -> str:
    so = os.path.join(ORACLE_DIR, "libefo_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".h")) or f == "Makefile"]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libefo_oracle.so"])
    return so


_BACKEND = None   # set by `with efo.backend(...)`: operator wrappers below then call another library


class _Proxy:
    """Routes efo_<op> to <prefix><op> of another library with the same signatures (oracle/_ref, no-FMA oracle)."""

    OPS = {"efo_" + n for n in ("pyr_down_u16", "create_vmap", "create_nmap", "transform_maps", "copy_maps", "resize_map",
                                "pyr_down_gauss_f", "pyr_down_uchar_gauss", "vertices_to_depth", "bgr_to_intensity",
                                "derivative_images", "project_to_point_cloud", "icp_step", "rgb_residual", "rgb_step", "so3_step")}

    MAP_OPS = {"efo_" + n for n in ("filter_depth", "metricise_depth", "seed_map", "predict_indices", "combined_predict", "synthesize_depth", "fill_in", "fuse", "clean_deform",
                                    "clean", "sample_graph", "resize_nearest")}

    # the tracking DRIVER (RGBDOdometry): handle-based, so an Odometry must be created and used under the same backend
    ODOM_OPS = {"efo_odom_" + n for n in ("create", "destroy", "init_icp", "init_icp_model", "init_icp_maps", "init_rgb_model", "init_rgb",
                                          "init_first_rgb", "track", "stats")}

    def __init__(self, so, prefix, default, ops=None):
        self._so, self._prefix, self._default = so, prefix, default
        self.OPS = ops if ops is not None else _Proxy.OPS

    def __getattr__(self, name):
        if name in self.OPS:   # the routed operators; everything else (handles, drivers) stays on the oracle
            return getattr(self._so, self._prefix + name[len("efo_"):])
        return getattr(self._default, name)


REF_SO = os.path.join(ORACLE_DIR, "_ref", "libefr_cuda.so")
REF_GLSL_SO = os.path.join(ORACLE_DIR, "_ref", "libefr_glsl.so")
REF_DRIVER_SO = os.path.join(ORACLE_DIR, "_ref", "libefr_driver.so")
REF_DRIVER_DSQRT_SO = os.path.join(ORACLE_DIR, "_ref", "libefr_driver_dsqrt.so")   # unqualified sqrt(float) read as ::sqrt(double)
NOFMA_SO = os.path.join(ORACLE_DIR, "libefo_oracle.so")        # reference rounding IS the default oracle since round 5 ("nofma" backends: kept as names)
FAST_SO = os.path.join(ORACLE_DIR, "libefo_oracle_fast.so")     # the opt-in fast build's specification: fused multiply-adds + the fast order


def have_reference() -> bool:
    """oracle/_ref/libefr_cuda.so = the reference's own Core/Cuda sources compiled for the CPU (oracle/Makefile `ref`).
    It can only be BUILT where /root/reference exists; the built file travels with the repo snapshot."""
    if not os.path.exists(REF_SO) and os.path.isdir("/root/reference/Core/Cuda"):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])
    return os.path.exists(REF_SO)


def have_reference_glsl() -> bool:
    """oracle/_ref/libefr_glsl.so = the reference's own Core/Shaders sources compiled for the CPU (oracle/Makefile `refglsl`)."""
    if not os.path.exists(REF_GLSL_SO) and os.path.isdir("/root/reference/Core/Shaders"):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "refglsl"])
    return os.path.exists(REF_GLSL_SO)


def have_reference_driver() -> bool:
    """oracle/_ref/libefr_driver.so = the reference's own Core/Utils/RGBDOdometry.cpp compiled for the CPU against
    oracle/host_on_cpu (Eigen / Sophus / Pangolin in miniature) over its own CUDA operators (oracle/Makefile `refdriver`)."""
    if not (os.path.exists(REF_DRIVER_SO) and os.path.exists(REF_DRIVER_DSQRT_SO)) and os.path.isdir("/root/reference/Core/Utils"):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "refdriver"])
    return os.path.exists(REF_DRIVER_SO)


def reference_glsl_lib():
    assert have_reference_glsl()
    so = C.CDLL(REF_GLSL_SO)
    so.efg_set_texel_snap.argtypes = [C.c_double]
    return so


class backend:
    """with efo.backend("reference") / efo.backend("nofma"): the tracking-operator wrappers of this module run on
    the compiled reference / on the oracle built with -DEFO_NO_FMA instead of the default oracle."""

    def __init__(self, which):
        self.which = which

    def __enter__(self):
        global _BACKEND
        default = lib()
        if self.which == "reference":
            assert have_reference(), "oracle/_ref/libefr_cuda.so is missing and cannot be built here"
            _BACKEND = _Proxy(C.CDLL(REF_SO), "efr_", default)
        elif self.which == "reference_glsl":
            _BACKEND = _Proxy(reference_glsl_lib(), "efg_", default, _Proxy.MAP_OPS)
        elif self.which == "nofma":             # (the default oracle since round 5; the name is kept for the tests that pin it against the reference)
            _BACKEND = _Proxy(C.CDLL(build()), "efo_", default, _Proxy.OPS | _Proxy.MAP_OPS)
        elif self.which == "fast":              # the opt-in fast build's specification
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libefo_oracle_fast.so"])
            _BACKEND = _Proxy(C.CDLL(FAST_SO), "efo_", default, _Proxy.OPS | _Proxy.MAP_OPS)
        elif self.which == "nofma_driver":      # the oracle's tracking driver in the no-FMA build (its operators are then no-FMA too)
            so = C.CDLL(build())
            so.efo_odom_create.restype = P
            _BACKEND = _Proxy(so, "efo_", default, _Proxy.ODOM_OPS)
        elif self.which == "reference_driver":  # the reference's own RGBDOdometry.cpp, compiled
            assert have_reference_driver(), "oracle/_ref/libefr_driver.so is missing and cannot be built here"
            so = C.CDLL(REF_DRIVER_SO)
            so.efd_odom_create.restype = P
            _BACKEND = _Proxy(so, "efd_", default, _Proxy.ODOM_OPS)
        elif self.which == "reference_driver_dsqrt":
            assert have_reference_driver()
            so = C.CDLL(REF_DRIVER_DSQRT_SO)
            so.efd_odom_create.restype = P
            _BACKEND = _Proxy(so, "efd_", default, _Proxy.ODOM_OPS)
        else:
            raise ValueError(self.which)
        return self

    def __exit__(self, *a):
        global _BACKEND
        _BACKEND = None


def _load(path):
    so = C.CDLL(path)
    so.efo_odom_create.restype = P
    so.efo_odom_buffer.restype = P
    so.efo_fusion_create.restype = P
    so.efo_fusion_buffer.restype = P
    so.efo_fusion_odometry.restype = P
    so.efo_se3_log_norm.restype = C.c_double
    so.efo_expf_spec.restype = c_f
    so.efo_expf_spec.argtypes = [c_f]
    return so


def lib():
    global _LIB
    if _BACKEND is not None:
        return _BACKEND
    if _LIB is None:
        _LIB = _load(build())
    return _LIB


class whole_library:
    """with efo.whole_library("fast"): EVERYTHING of this module (Fusion objects included) runs on the oracle of the opt-in fast build
    (fused multiply-adds + the fast summation order: libefo_oracle_fast.so); "nofma" = the default oracle (reference rounding: the
    arithmetic the reference's own sources compute when compiled without contraction, oracle/README.md), kept as a name.
    Objects must be created and destroyed inside the block."""

    def __init__(self, which):
        assert which in ("nofma", "fast"), which
        self.which = which

    def __enter__(self):
        global _LIB
        self._saved = lib()
        if self.which == "fast":
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libefo_oracle_fast.so"])
            _LIB = _load(FAST_SO)
        return self

    def __exit__(self, *a):
        global _LIB
        _LIB = self._saved


def covariance(lastA):
    c = np.zeros(36, np.float64)
    lib().efo_covariance(ptr(np.ascontiguousarray(lastA, np.float64).reshape(36)), ptr(c))
    return c.reshape(6, 6)


def set_threads(n: int):
    """host threads for the parallel loops of the oracle (cpu_baseline "all cores" leg); results do not depend on it"""
    lib().efo_set_threads(c_i(int(n)))


def ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(P)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Cam(C.Structure):
    _fields_ = [("cols", c_i), ("rows", c_i), ("fx", c_f), ("fy", c_f), ("cx", c_f), ("cy", c_f)]


class FusionParams(C.Structure):
    _fields_ = [("width", c_i), ("height", c_i), ("fx", c_f), ("fy", c_f), ("cx", c_f), ("cy", c_f),
                ("timeDelta", c_i), ("confidence", c_f), ("depthCut", c_f), ("icpWeight", c_f),
                ("fastOdom", c_i), ("so3", c_i), ("frameToFrameRGB", c_i), ("pyramid", c_i), ("rgbOnly", c_i),
                ("maxSurfels", c_i)]


DATATERM = np.dtype([("zero", np.int16, 2), ("one", np.int16, 2), ("diff", np.float32), ("valid", np.uint8),
                     ("pad", np.uint8, 3)])
assert DATATERM.itemsize == 16

# ------------------------------------------------------------------------------------------------
# operator wrappers (numpy in / numpy out)
# ------------------------------------------------------------------------------------------------


def pyr_down_u16(src):
    h, w = src.shape
    dst = np.zeros((h // 2, w // 2), np.uint16)
    lib().efo_pyr_down_u16(ptr(src), c_i(w), c_i(h), ptr(dst))
    return dst


def create_vmap(depth, fx, fy, cx, cy, cutoff, vmap=None):
    h, w = depth.shape
    vmap = np.zeros((3 * h, w), np.float32) if vmap is None else vmap
    lib().efo_create_vmap(ptr(depth), c_i(w), c_i(h), c_f(fx), c_f(fy), c_f(cx), c_f(cy), c_f(cutoff), ptr(vmap))
    return vmap


def create_nmap(vmap, nmap=None):
    h3, w = vmap.shape
    nmap = np.zeros((h3, w), np.float32) if nmap is None else nmap
    lib().efo_create_nmap(ptr(vmap), c_i(w), c_i(h3 // 3), ptr(nmap))
    return nmap


def transform_maps(vmap, nmap, R, t):
    v, n = vmap.copy(), nmap.copy()
    lib().efo_transform_maps(ptr(v), ptr(n), c_i(v.shape[1]), c_i(v.shape[0] // 3), ptr(f32(R).reshape(9)), ptr(f32(t)))
    return v, n


def copy_maps(vtex, ntex):
    h, w, _ = vtex.shape
    tmp = np.zeros((h, w, 4), np.float32)
    vm = np.zeros((3 * h, w), np.float32)
    nm = np.zeros((3 * h, w), np.float32)
    lib().efo_copy_maps(ptr(vtex), ptr(ntex), c_i(w), c_i(h), ptr(tmp), ptr(vm), ptr(nm))
    return tmp, vm, nm


def resize_map(src, normalize, out=None):
    h3, w = src.shape
    out = np.zeros((h3 // 2, w // 2), np.float32) if out is None else out
    lib().efo_resize_map(ptr(src), c_i(w), c_i(h3 // 3), ptr(out), c_i(int(normalize)))
    return out


def pyr_down_gauss_f(src):
    h, w = src.shape
    dst = np.zeros((h // 2, w // 2), np.float32)
    lib().efo_pyr_down_gauss_f(ptr(src), c_i(w), c_i(h), ptr(dst))
    return dst


def pyr_down_uchar_gauss(src):
    h, w = src.shape
    dst = np.zeros((h // 2, w // 2), np.uint8)
    lib().efo_pyr_down_uchar_gauss(ptr(src), c_i(w), c_i(h), ptr(dst))
    return dst


def vertices_to_depth(vmaps_tmp, cutoff):
    h, w, _ = vmaps_tmp.shape
    dst = np.zeros((h, w), np.float32)
    lib().efo_vertices_to_depth(ptr(vmaps_tmp), c_i(w), c_i(h), c_f(cutoff), ptr(dst))
    return dst


def bgr_to_intensity(rgba):
    h, w, _ = rgba.shape
    dst = np.zeros((h, w), np.uint8)
    lib().efo_bgr_to_intensity(ptr(rgba), c_i(w), c_i(h), ptr(dst))
    return dst


def derivative_images(img):
    h, w = img.shape
    dx = np.zeros((h, w), np.int16)
    dy = np.zeros((h, w), np.int16)
    lib().efo_derivative_images(ptr(img), c_i(w), c_i(h), ptr(dx), ptr(dy))
    return dx, dy


def project_to_point_cloud(depth, fx, fy, cx, cy):
    h, w = depth.shape
    cloud = np.zeros((h, w, 3), np.float32)
    lib().efo_project_to_point_cloud(ptr(depth), c_i(w), c_i(h), c_f(fx), c_f(fy), c_f(cx), c_f(cy), ptr(cloud))
    return cloud


def icp_step(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, distThres, angleThres):
    h3, w = vmap_curr.shape
    A = np.zeros((6, 6), np.float32)
    b = np.zeros(6, np.float32)
    res = np.zeros(2, np.float32)
    fx, fy, cx, cy = intr
    lib().efo_icp_step(ptr(f32(Rcurr).reshape(9)), ptr(f32(tcurr)), ptr(vmap_curr), ptr(nmap_curr),
                       ptr(f32(Rprev_inv).reshape(9)), ptr(f32(tprev)), c_f(fx), c_f(fy), c_f(cx), c_f(cy),
                       ptr(vmap_g_prev), ptr(nmap_g_prev), c_f(distThres), c_f(angleThres), c_i(w), c_i(h3 // 3),
                       ptr(A), ptr(b), ptr(res))
    return A, b, res


def rgb_residual(minScale, dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage, maxDepthDelta, kt, krkinv):
    h, w = nextImage.shape
    corres = np.zeros((h, w), DATATERM)
    sigma = c_i(0)
    count = c_i(0)
    lib().efo_rgb_residual(c_f(minScale), ptr(dIdx), ptr(dIdy), ptr(lastDepth), ptr(nextDepth), ptr(lastImage),
                           ptr(nextImage), ptr(corres), c_f(maxDepthDelta), ptr(f32(kt)), ptr(f32(krkinv).reshape(9)),
                           c_i(w), c_i(h), C.byref(sigma), C.byref(count))
    return corres, sigma.value, count.value


def rgb_step(corres, sigma, cloud, fx, fy, dIdx, dIdy, sobelScale):
    h, w = corres.shape
    A = np.zeros((6, 6), np.float32)
    b = np.zeros(6, np.float32)
    lib().efo_rgb_step(ptr(corres), c_f(sigma), ptr(cloud), c_f(fx), c_f(fy), ptr(dIdx), ptr(dIdy), c_f(sobelScale),
                       c_i(w), c_i(h), ptr(A), ptr(b))
    return A, b


def so3_step(lastImage, nextImage, imageBasis, kinv, krlr):
    h, w = nextImage.shape
    A = np.zeros((3, 3), np.float32)
    b = np.zeros(3, np.float32)
    res = np.zeros(2, np.float32)
    lib().efo_so3_step(ptr(lastImage), ptr(nextImage), ptr(f32(imageBasis).reshape(9)), ptr(f32(kinv).reshape(9)),
                       ptr(f32(krlr).reshape(9)), c_i(w), c_i(h), ptr(A), ptr(b), ptr(res))
    return A, b, res


def filter_depth(raw, maxD):
    h, w = raw.shape
    out = np.zeros((h, w), np.uint16)
    lib().efo_filter_depth(ptr(raw), c_i(w), c_i(h), c_f(maxD), ptr(out))
    return out


def metricise_depth(d, maxD):
    h, w = d.shape
    out = np.zeros((h, w), np.float32)
    lib().efo_metricise_depth(ptr(d), c_i(w), c_i(h), c_f(maxD), ptr(out))
    return out


def make_cam(w, h, fx, fy, cx, cy):
    return Cam(w, h, fx, fy, cx, cy)


def seed_map(cam, rgb, dm, dmf, time, maxDepth):
    out = np.zeros((cam.cols * cam.rows, 12), np.float32)
    n = lib().efo_seed_map(C.byref(cam), ptr(rgb), ptr(dm), ptr(dmf), c_i(time), c_f(maxDepth), ptr(out))
    return out[:n].copy()


def _T(T):
    return np.ascontiguousarray(T, dtype=np.float64).reshape(16)


def predict_indices(cam, T_wc, time, surfels, maxDepth, timeDelta):
    Pn = cam.cols * cam.rows
    idx = np.zeros((cam.rows, cam.cols), np.uint32)
    vc = np.zeros((cam.rows, cam.cols, 4), np.float32)
    ct = np.zeros((cam.rows, cam.cols, 4), np.float32)
    nr = np.zeros((cam.rows, cam.cols, 4), np.float32)
    s = f32(surfels)
    lib().efo_predict_indices(C.byref(cam), ptr(_T(T_wc)), c_i(time), ptr(s), c_i(len(s)), c_f(maxDepth), c_i(timeDelta),
                              ptr(idx), ptr(vc), ptr(ct), ptr(nr))
    return idx, vc, ct, nr


def combined_predict(cam, T_wc, surfels, maxDepth, confThreshold, time, maxTime, timeDelta):
    img = np.zeros((cam.rows, cam.cols, 4), np.uint8)
    vt = np.zeros((cam.rows, cam.cols, 4), np.float32)
    nm = np.zeros((cam.rows, cam.cols, 4), np.float32)
    tm = np.zeros((cam.rows, cam.cols), np.uint16)
    s = f32(surfels)
    lib().efo_combined_predict(C.byref(cam), ptr(_T(T_wc)), ptr(s), c_i(len(s)), c_f(maxDepth), c_f(confThreshold),
                               c_i(time), c_i(maxTime), c_i(timeDelta), ptr(img), ptr(vt), ptr(nm), ptr(tm))
    return img, vt, nm, tm


def synthesize_depth(cam, T_wc, surfels, maxDepth, confThreshold, time, maxTime, timeDelta):
    d = np.zeros((cam.rows, cam.cols), np.float32)
    s = f32(surfels)
    lib().efo_synthesize_depth(C.byref(cam), ptr(_T(T_wc)), ptr(s), c_i(len(s)), c_f(maxDepth), c_f(confThreshold), c_i(time),
                               c_i(maxTime), c_i(timeDelta), ptr(d))
    return d


def fill_in(cam, image, vertex, normal, depthFiltered, rgb, passthrough=0, passthroughImage=0):
    fi = np.zeros_like(image)
    fv = np.zeros_like(vertex)
    fn = np.zeros_like(normal)
    lib().efo_fill_in(C.byref(cam), ptr(image), ptr(vertex), ptr(normal), ptr(depthFiltered), ptr(rgb),
                      c_i(passthrough), c_i(passthroughImage), ptr(fi), ptr(fv), ptr(fn))
    return fi, fv, fn


def dense_enough(cam, image):
    return bool(lib().efo_dense_enough(C.byref(cam), ptr(image)))


def fuse(cam, T_wc, time, rgb, dm, dmf, idx, vc, ct, nr, maxDepth, weighting, surfels):
    s = f32(surfels).copy()
    newu = np.zeros((cam.cols * cam.rows, 12), np.float32)
    n = lib().efo_fuse(C.byref(cam), ptr(_T(T_wc)), c_i(time), ptr(rgb), ptr(dm), ptr(dmf), ptr(idx), ptr(vc), ptr(ct),
                       ptr(nr), c_f(maxDepth), c_f(weighting), ptr(s), c_i(len(s)), ptr(newu))
    return s, newu[:n].copy()


def clean(cam, T_wc, time, idx, vc, ct, nr, confThreshold, timeDelta, maxDepth, surfels, newUnstable):
    s = f32(surfels)
    nu = f32(newUnstable).reshape(-1, 12)
    out = np.zeros((len(s) + len(nu), 12), np.float32)
    n = lib().efo_clean(C.byref(cam), ptr(_T(T_wc)), c_i(time), ptr(idx), ptr(vc), ptr(ct), ptr(nr), c_f(confThreshold),
                        c_i(timeDelta), c_f(maxDepth), ptr(s), c_i(len(s)), ptr(nu), c_i(len(nu)), ptr(out))
    return out[:n].copy()


def clean_deform(cam, T_wc, time, idx, vc, ct, nr, confThreshold, timeDelta, maxDepth, surfels, newUnstable, graph, depth, isFern=0):
    """clean with the deformation graph applied: graph = (nodes, 16) float32 sorted by time, depth = synthesizeDepth image"""
    s = f32(surfels)
    nu = f32(newUnstable).reshape(-1, 12)
    g = f32(graph).reshape(-1, 16)
    d = f32(depth)
    out = np.zeros((len(s) + len(nu), 12), np.float32)
    n = lib().efo_clean_deform(C.byref(cam), ptr(_T(T_wc)), c_i(time), ptr(idx), ptr(vc), ptr(ct), ptr(nr), c_f(confThreshold),
                               c_i(timeDelta), c_f(maxDepth), ptr(s), c_i(len(s)), ptr(nu), c_i(len(nu)), ptr(g), c_i(len(g)), ptr(d),
                               c_i(isFern), ptr(out))
    return out[:n].copy()


# ------------------------------------------------------------------------------------------------
# tracking driver + whole-frame objects
# ------------------------------------------------------------------------------------------------
class Odometry:
    BUF = dict(vmap_curr=(0, np.float32, 3), nmap_curr=(1, np.float32, 3), vmap_g_prev=(2, np.float32, 3),
               nmap_g_prev=(3, np.float32, 3), lastDepth=(4, np.float32, 1), nextDepth=(5, np.float32, 1),
               lastImage=(6, np.uint8, 1), nextImage=(7, np.uint8, 1), lastNextImage=(8, np.uint8, 1),
               dIdx=(9, np.int16, 1), dIdy=(10, np.int16, 1), depth_tmp=(11, np.uint16, 1))

    def __init__(self, w, h, cx, cy, fx, fy, handle=None):
        self.w, self.h = w, h
        self.own = handle is None
        self._l = lib()   # the backend this object lives on (an Odometry is created and used on ONE library)
        self.h_ = P(self._l.efo_odom_create(c_i(w), c_i(h), c_f(cx), c_f(cy), c_f(fx), c_f(fy))) if handle is None else P(handle)

    def __del__(self):
        if getattr(self, "own", False) and self.h_:
            self._l.efo_odom_destroy(self.h_)
            self.h_ = None

    def init_icp(self, depth_filtered, cutoff):
        self._l.efo_odom_init_icp(self.h_, ptr(depth_filtered), c_f(cutoff))

    def init_icp_model(self, vtex, ntex, T_wc):
        self._l.efo_odom_init_icp_model(self.h_, ptr(f32(vtex)), ptr(f32(ntex)), ptr(_T(T_wc)))

    def init_icp_maps(self, vtex, ntex):
        self._l.efo_odom_init_icp_maps(self.h_, ptr(f32(vtex)), ptr(f32(ntex)))

    def init_rgb_model(self, rgba):
        self._l.efo_odom_init_rgb_model(self.h_, ptr(rgba))

    def init_rgb(self, rgba):
        self._l.efo_odom_init_rgb(self.h_, ptr(rgba))

    def init_first_rgb(self, rgba):
        self._l.efo_odom_init_first_rgb(self.h_, ptr(rgba))

    def track(self, T_wc, rgbOnly=False, icpWeight=10.0, pyramid=True, fastOdom=False, so3=True):
        T = _T(T_wc).copy()
        self._l.efo_odom_track(self.h_, ptr(T), c_i(int(rgbOnly)), c_f(icpWeight), c_i(int(pyramid)), c_i(int(fastOdom)), c_i(int(so3)))
        return T.reshape(4, 4)

    def stats(self):
        out = np.zeros(6, np.float32)
        A = np.zeros((6, 6), np.float64)
        b = np.zeros(6, np.float64)
        self._l.efo_odom_stats(self.h_, ptr(out), ptr(A), ptr(b))
        return out, A, b

    def buffer(self, name, level=0):
        which, dt, planes = self.BUF[name]
        w, h = self.w >> level, self.h >> level
        addr = self._l.efo_odom_buffer(self.h_, c_i(which), c_i(level))
        n = w * h * planes
        arr = np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,))
        return arr.view(dt).reshape(h * planes, w).copy()


def resize_nearest(img, factor):
    """Resize::{image,vertex,time}: img [H, W] or [H, W, C] of any dtype -> [H // factor, W // factor(, C)]"""
    a = np.ascontiguousarray(img)
    h, w = a.shape[:2]
    elem = a.dtype.itemsize * (a.shape[2] if a.ndim == 3 else 1)
    out = np.zeros((h // factor, w // factor) + a.shape[2:], a.dtype)
    lib().efo_resize_nearest(ptr(a), c_i(w), c_i(h), c_i(elem), c_i(factor), ptr(out))
    return out


def sample_graph(surfels):
    s = f32(surfels).reshape(-1, 12)
    out = np.zeros((len(s) // 5000 + 1, 4), np.float32)
    fn = getattr(lib(), "efo_sample_graph")
    n = fn(ptr(s), c_i(len(s)), ptr(out))
    return out[:n].copy()


class LocalLoop(C.Structure):
    _fields_ = [("attempted", c_i), ("cov_ok", c_i), ("gates_ok", c_i), ("n_constraints", c_i), ("applied", c_i),
                ("graph_nodes", c_i), ("stats", c_f * 6), ("cov_diag", C.c_double * 6), ("T_wc_curr", C.c_double * 16),
                ("T_wc_est", C.c_double * 16)]


LOOP_SOLVER = C.CFUNCTYPE(c_i, C.c_void_p, C.POINTER(LocalLoop), C.POINTER(C.c_double), c_i, C.POINTER(c_f), C.POINTER(c_i))


class GlobalLoop(C.Structure):
    _fields_ = [("attempted", c_i), ("closest", c_i), ("n_constraints", c_i), ("accepted", c_i), ("graph_nodes", c_i), ("icp_error", c_f),
                ("icp_count", c_f), ("T_wc_recovery", C.c_double * 16)]


DEFORM_SOLVER = C.CFUNCTYPE(c_i, C.c_void_p, c_i, C.POINTER(C.c_double), c_i, C.POINTER(C.c_double), C.POINTER(C.c_int64), c_i, C.POINTER(c_f),
                            C.POINTER(c_i), C.POINTER(C.c_double), C.POINTER(c_i))


class Fusion:
    def __init__(self, **kw):
        self.p = FusionParams()
        lib().efo_fusion_default_params(C.byref(self.p))
        for k, v in kw.items():
            assert hasattr(self.p, k), k
            setattr(self.p, k, v)
        self.h_ = P(lib().efo_fusion_create(C.byref(self.p)))

    def __del__(self):
        if getattr(self, "h_", None):
            lib().efo_fusion_destroy(self.h_)
            self.h_ = None

    def process_frame(self, rgb, depth, ts=0, weight=1.0, T_wc=None):
        lib().efo_fusion_process_frame(self.h_, ptr(rgb), ptr(depth), C.c_int64(ts), c_f(weight),
                                       ptr(_T(T_wc)) if T_wc is not None else None)

    def set_deformation(self, graph, isFern=False):
        g = f32(graph).reshape(-1, 16)
        lib().efo_fusion_set_deformation(self.h_, ptr(g), c_i(len(g)), c_i(int(isFern)))

    # ---- local loop closure, front half (ElasticFusion.cpp:447-511) ----
    def set_close_loops(self, on=True, icpCountThresh=35000, icpErrThresh=5e-05, covThresh=1e-05):
        lib().efo_fusion_set_close_loops(self.h_, c_i(int(on)), c_i(icpCountThresh), c_f(icpErrThresh), c_f(covThresh))

    def set_loop_solver(self, fn):
        """fn(info: LocalLoop, constraints [n, 8] float64) -> None | graph [nodes, 16] float32 (accepted)."""
        if fn is None:
            self._solver = None
            lib().efo_fusion_set_loop_solver(self.h_, None, None)
            return

        def tramp(user, info, cons, n, graph_out, nodes_out):
            c = np.ctypeslib.as_array(cons, shape=(n, 8)).copy() if n > 0 else np.zeros((0, 8))
            g = fn(info.contents, c)
            if g is None:
                return 0
            g = f32(g).reshape(-1, 16)
            C.memmove(graph_out, g.ctypes.data, g.nbytes)
            nodes_out[0] = len(g)
            return 1
        self._solver = LOOP_SOLVER(tramp)
        lib().efo_fusion_set_loop_solver(self.h_, self._solver, None)

    # ---- global loop closure (ElasticFusion.cpp:392-445,609-618) ----
    def set_tick(self, tick):
        lib().efo_fusion_set_tick(self.h_, c_i(int(tick)))

    def pose_qt(self):
        qt = np.zeros(7, np.float64)
        lib().efo_fusion_get_pose_qt(self.h_, ptr(qt))
        return qt

    def checkpoint(self, last_rgb, last_depth):
        """what a replay carries from one process_frame to the next (api.ElasticFusion.checkpoint's fields)"""
        return dict(map=self.map().copy(), tick=self.tick(), qt=self.pose_qt(), rgb=np.ascontiguousarray(last_rgb, np.uint8).copy(),
                    depth=np.ascontiguousarray(last_depth, np.uint16).copy())

    def restore(self, ck):
        """the oracle's side of api.ElasticFusion.restore(checkpoint): dict(map [n, 12] float32, tick, qt [7] float64, rgb, depth)"""
        m = f32(ck["map"]).reshape(-1, 12)
        assert len(m) <= self.p.maxSurfels, (len(m), self.p.maxSurfels)
        qt = np.ascontiguousarray(ck["qt"], np.float64).reshape(7)
        rgb, depth = np.ascontiguousarray(ck["rgb"], np.uint8), np.ascontiguousarray(ck["depth"], np.uint16)
        lib().efo_fusion_restore(self.h_, ptr(m), c_i(len(m)), c_i(int(ck["tick"])), ptr(qt), ptr(rgb), ptr(depth))

    # ---- relocalisation (ElasticFusion.cpp:326-366,411-413,536,601-604,624-649) ----
    def set_reloc(self, on=True):
        lib().efo_fusion_set_reloc(self.h_, c_i(int(on)))

    def reloc_state(self):
        """dict(lost, trackingOk, trackingCount, lastFrameRecovery) after the last frame"""
        out = np.zeros(4, np.int32)
        lib().efo_fusion_reloc_state(self.h_, ptr(out))
        return dict(lost=bool(out[0]), trackingOk=bool(out[1]), trackingCount=int(out[2]), lastFrameRecovery=bool(out[3]))

    def enable_ferns(self, num=500, photoThresh=115.0, fernThresh=0.3095, seed=0):
        lib().efo_fusion_enable_ferns(self.h_, c_i(num), c_f(photoThresh), c_f(fernThresh), C.c_uint(seed))

    def ferns(self):
        """the instance's fern database behind the interface of elasticfusion_amd.api.Ferns (not owned)"""
        lib().efo_fusion_ferns.restype = C.c_void_p
        f = Ferns.__new__(_oracle_ferns_class())
        f._L, f._h = lib(), P(lib().efo_fusion_ferns(self.h_))
        f.num, f.w, f.h = 500, self.p.width // 8, self.p.height // 8
        for name, res in (("count", c_i), ("last_closest", c_i), ("block_hd_aware", c_f), ("photometric_check", c_f)):
            f._f(name).restype = res
        f._f("count").argtypes = f._f("last_closest").argtypes = [P]
        f._f("get_frame").argtypes = [P, c_i, P, P, P, P, P, P, P]
        f._f("get_table").argtypes = [P, P]
        f._owned = False
        return f

    def fern_view(self, which):
        """(rgba, verts, norms) at 1/8 resolution: which = 0 mid-frame (findFrame), 1 end of frame (addFrame); None if that step did not run"""
        h, w = self.p.height // 8, self.p.width // 8
        img, v, n = np.zeros((h, w, 4), np.uint8), np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
        return (img, v, n) if lib().efo_fusion_fern_view(self.h_, c_i(which), ptr(img), ptr(v), ptr(n)) else None

    def set_deform_solver(self, fn):
        """fn(fernMatch, rows [n, 10], poses [k, 4, 4], pose_times [k]) -> None | dict(graph [nodes, 16], poses [k, 4, 4], new_relative rows [m, 10])"""
        def tramp(user, fernMatch, rows, n, poses, times, k, graph_out, nodes_out, rel_out, n_rel):
            r = np.ctypeslib.as_array(rows, shape=(n, 10)).copy() if n > 0 else np.zeros((0, 10))
            Pz = np.ctypeslib.as_array(poses, shape=(k, 4, 4)).copy() if k > 0 else np.zeros((0, 4, 4))
            tz = np.ctypeslib.as_array(times, shape=(k,)).copy() if k > 0 else np.zeros(0, np.int64)
            out = fn(bool(fernMatch), r, Pz, tz)
            if out is None:
                return 0
            g = f32(out["graph"]).reshape(-1, 16)
            C.memmove(graph_out, g.ctypes.data, g.nbytes)
            nodes_out[0] = len(g)
            if k > 0:
                q = np.ascontiguousarray(out["poses"], np.float64).reshape(k, 16)
                C.memmove(poses, q.ctypes.data, q.nbytes)
            if rel_out:
                rel = np.ascontiguousarray(out.get("new_relative", np.zeros((0, 10))), np.float64).reshape(-1, 10)
                if len(rel):
                    C.memmove(rel_out, rel.ctypes.data, rel.nbytes)
                n_rel[0] = len(rel)
            return 1
        self._deform = DEFORM_SOLVER(tramp)
        lib().efo_fusion_set_deform_solver(self.h_, self._deform, None)

    def global_loop(self):
        info = GlobalLoop()
        lib().efo_fusion_global_loop(self.h_, C.byref(info))
        return info

    def relative_constraints(self):
        rows = np.zeros((4096, 10), np.float64)
        n = lib().efo_fusion_relative_constraints(self.h_, ptr(rows), c_i(len(rows)))
        return rows[:n].copy()

    def trajectory(self):
        n = lib().efo_fusion_trajectory(self.h_, None, c_i(0))
        T = np.zeros((max(n, 1), 4, 4), np.float64)
        lib().efo_fusion_trajectory(self.h_, ptr(T), c_i(n))
        return T[:n]

    def local_loop(self):
        info = LocalLoop()
        cons = np.zeros((4096, 8), np.float64)
        lib().efo_fusion_local_loop.restype = c_i
        n = lib().efo_fusion_local_loop(self.h_, C.byref(info), ptr(cons), c_i(len(cons)))
        return info, cons[:n].copy()

    def old_buffer(self, name):
        which = dict(image=0, vertex=1, normal=2, time=3)[name]
        dt, ch = self._BUF[which]
        lib().efo_fusion_old_buffer.restype = C.c_void_p
        addr = lib().efo_fusion_old_buffer(self.h_, c_i(which))
        n = self.p.width * self.p.height * ch
        a = np.ctypeslib.as_array(C.cast(addr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()
        return a.reshape(self.p.height, self.p.width, ch) if ch > 1 else a.reshape(self.p.height, self.p.width)

    def pose(self):
        T = np.zeros(16, np.float64)
        lib().efo_fusion_get_pose(self.h_, ptr(T))
        return T.reshape(4, 4)

    def map_count(self):
        return lib().efo_fusion_map_count(self.h_)

    def map(self):
        n = self.map_count()
        out = np.zeros((n, 12), np.float32)
        lib().efo_fusion_map_download(self.h_, ptr(out))
        return out

    def map_reference(self):
        """GlobalModel::downloadMap as the reference has it: the pre-clean buffer truncated to the post-clean count (quirk Q14)"""
        out = np.zeros((self.map_count(), 12), np.float32)
        lib().efo_fusion_map_download_reference(self.h_, ptr(out))
        return out

    def tick(self):
        return lib().efo_fusion_tick(self.h_)

    def stats(self):
        out = np.zeros(6, np.float32)
        lib().efo_fusion_stats(self.h_, ptr(out))
        return out

    def odometry(self):
        return Odometry(self.p.width, self.p.height, self.p.cx, self.p.cy, self.p.fx, self.p.fy,
                        handle=lib().efo_fusion_odometry(self.h_))

    _BUF = {0: (np.uint8, 4), 1: (np.float32, 4), 2: (np.float32, 4), 3: (np.uint16, 1), 4: (np.uint8, 4),
            5: (np.float32, 4), 6: (np.float32, 4), 7: (np.uint32, 1), 8: (np.float32, 4), 9: (np.float32, 4),
            10: (np.float32, 4), 11: (np.uint16, 1), 12: (np.float32, 1), 13: (np.float32, 1)}
    NAMES = dict(image=0, vertex=1, normal=2, time=3, fill_image=4, fill_vertex=5, fill_normal=6, index=7,
                 vertConf=8, colorTime=9, normRad=10, depthFiltered=11, depthMetric=12, depthMetricFiltered=13)

    def buffer(self, name):
        which = self.NAMES[name]
        dt, ch = self._BUF[which]
        addr = lib().efo_fusion_buffer(self.h_, c_i(which))
        n = self.p.width * self.p.height * ch
        arr = np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,))
        a = arr.view(dt).copy()
        return a.reshape(self.p.height, self.p.width, ch) if ch > 1 else a.reshape(self.p.height, self.p.width)


_ORACLE_FERNS = None


def _oracle_ferns_class():
    global _ORACLE_FERNS
    if _ORACLE_FERNS is None:
        from elasticfusion_amd import api

        class OracleFerns(api.Ferns):
            _prefix = "efo_ferns_"

            def _library(self):
                return lib()

        _ORACLE_FERNS = OracleFerns
    return _ORACLE_FERNS


class Ferns:
    """the oracle's restatement of the fern database (oracle/efo_ferns.cpp) behind the interface of elasticfusion_amd.api.Ferns"""

    def __new__(cls, *a, **kw):
        if cls is Ferns:
            return _oracle_ferns_class()(*a, **kw)
        return object.__new__(cls)
