"""Shared harness of the tracking-operator parity tests: one fixed set of inputs taken from a real tracking state, and
one function that pushes them through the 16 operators of the reference's Core/Cuda/cudafuncs.cuh:61-169 on any
backend with the oracle's Python signature (tests/efo.py):

    efo                         the CPU oracle (FMA specification)            oracle/libefo_oracle.so
    efo under backend("nofma")  the same restatement without fused mul-adds   oracle/libefo_oracle_nofma.so
    efo under backend("reference")  the REFERENCE's own sources on the CPU    oracle/_ref/libefr_cuda.so
    HipOps(api.ops)             the HIP kernels through the C ABI             elasticfusion_amd/libefusion_hip[_nofma].so

TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import numpy as np

FX, FY, CX, CY = 528.0, 528.0, 320.0, 240.0


def lvl_intr(level):
    d = 1 << level
    return FX / d, FY / d, CX / d, CY / d


def rot(rx, ry, rz):
    cx_, sx, cy_, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx_, -sx], [0, sx, cx_]])
    Ry = np.array([[cy_, 0, sy], [0, 1, 0], [-sy, 0, cy_]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).astype(np.float32)


def make_inputs(fusion, frame_rgb, level):
    """Inputs of all operators at pyramid level `level` from an oracle Fusion object positioned after a few frames."""
    odo = fusion.odometry()
    s = 1 << level
    rgba = np.full(frame_rgb.shape[:2] + (4,), 255, np.uint8)
    rgba[..., :3] = frame_rgb
    T = fusion.pose()
    Rprev = T[:3, :3].astype(np.float32)
    tprev = T[:3, 3].astype(np.float32)
    fx, fy, cx, cy = lvl_intr(level)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
    f2, _, c2x, c2y = lvl_intr(2)
    K2 = np.array([[f2, 0, c2x], [0, f2, c2y], [0, 0, 1]], np.float64)
    R2 = rot(0.003, -0.004, 0.002).astype(np.float64)
    th = 0.05
    inp = dict(
        level=np.int32(level),
        depth_u16=odo.buffer("depth_tmp", level),
        vtex=np.ascontiguousarray(fusion.buffer("fill_vertex")[::s, ::s]),
        ntex=np.ascontiguousarray(fusion.buffer("fill_normal")[::s, ::s]),
        rgba=np.ascontiguousarray(rgba[::s, ::s]),
        xf_R=np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float32),
        xf_t=np.array([0.01, -0.02, 0.03], np.float32),
        vmap_curr=odo.buffer("vmap_curr", level), nmap_curr=odo.buffer("nmap_curr", level),
        vmap_g_prev=odo.buffer("vmap_g_prev", level), nmap_g_prev=odo.buffer("nmap_g_prev", level),
        Rprev_inv=np.linalg.inv(Rprev).astype(np.float32), tprev=tprev,
        Rcurr=(Rprev @ rot(0.004, -0.003, 0.002)).astype(np.float32),
        tcurr=tprev + np.array([0.003, -0.002, 0.004], np.float32),
        dIdx=odo.buffer("dIdx", level), dIdy=odo.buffer("dIdy", level),
        lastDepth=odo.buffer("lastDepth", level), nextDepth=odo.buffer("nextDepth", level),
        lastImage=odo.buffer("lastImage", level), nextImage=odo.buffer("nextImage", level),
        kt=np.array([0.4, -0.3, 0.002], np.float32) / s,
        krkinv=(K @ rot(0.002, -0.001, 0.001).astype(np.float64) @ np.linalg.inv(K)).astype(np.float32),
        so3_last=odo.buffer("lastNextImage", 2), so3_next=odo.buffer("nextImage", 2),
        so3_basis=(K2 @ R2 @ np.linalg.inv(K2)).astype(np.float32), so3_kinv=np.linalg.inv(K2).astype(np.float32),
        so3_krlr=(K2 @ R2).astype(np.float32),
    )
    return {k: np.ascontiguousarray(v) for k, v in inp.items()}


def run_ops(be, inp):
    """All 16 operators on backend `be`; returns {name: array}.  Invalid correspondences are blanked (the reference
    leaves their `zero`/`one`/`diff` fields unwritten)."""
    level = int(np.asarray(inp["level"]).reshape(-1)[0])
    fx, fy, cx, cy = lvl_intr(level)
    g = (5, 3, 1)[level]
    out = {}
    out["pyr_down_u16"] = be.pyr_down_u16(inp["depth_u16"])
    v = be.create_vmap(inp["depth_u16"], fx, fy, cx, cy, 20.0)
    out["create_vmap"] = v
    out["create_nmap"] = be.create_nmap(v)
    tmp, vm, nm = be.copy_maps(inp["vtex"], inp["ntex"])
    out["copy_maps_tmp"], out["copy_maps_v"], out["copy_maps_n"] = tmp, vm, nm
    v1, n1 = be.resize_map(vm, False), be.resize_map(nm, True)
    out["resize_vmap"], out["resize_nmap"] = v1, n1
    out["transform_v"], out["transform_n"] = be.transform_maps(v1, n1, inp["xf_R"], inp["xf_t"])
    d = be.vertices_to_depth(tmp, 6.0)
    out["vertices_to_depth"] = d
    out["pyr_down_gauss_f"] = be.pyr_down_gauss_f(d)
    i0 = be.bgr_to_intensity(inp["rgba"])
    out["bgr_to_intensity"] = i0
    out["sobel_dx"], out["sobel_dy"] = be.derivative_images(i0)
    out["pyr_down_uchar_gauss"] = be.pyr_down_uchar_gauss(i0)
    cloud = be.project_to_point_cloud(inp["lastDepth"], fx, fy, cx, cy)
    out["point_cloud"] = cloud
    A, b, res = be.icp_step(inp["Rcurr"], inp["tcurr"], inp["vmap_curr"], inp["nmap_curr"], inp["Rprev_inv"], inp["tprev"],
                            (fx, fy, cx, cy), inp["vmap_g_prev"], inp["nmap_g_prev"], 0.10, float(np.sin(20 * 3.14159254 / 180)))
    out["icp_A"], out["icp_b"], out["icp_res"] = A, b, res
    c, sig, cnt = be.rgb_residual(float(g * g * 64), inp["dIdx"], inp["dIdy"], inp["lastDepth"], inp["nextDepth"], inp["lastImage"],
                                  inp["nextImage"], 0.07, inp["kt"], inp["krkinv"])
    valid = c["valid"] != 0
    out["residual_sums"] = np.array([sig, cnt], np.int64)
    out["residual_valid"] = valid.astype(np.uint8)
    for f in ("zero", "one", "diff"):
        x = c[f].copy()
        x[~valid] = 0
        out["residual_" + f] = x
    A, b = be.rgb_step(c, float(np.sqrt(cnt)), cloud, fx, fy, inp["dIdx"], inp["dIdy"], 0.125)
    out["rgb_A"], out["rgb_b"] = A, b
    out["so3_A"], out["so3_b"], out["so3_res"] = be.so3_step(inp["so3_last"], inp["so3_next"], inp["so3_basis"], inp["so3_kinv"],
                                                             inp["so3_krlr"])
    return out


class HipOps:
    """api.ops with the oracle's names."""

    def __init__(self, ops):
        self._o = ops

    def __getattr__(self, name):
        return getattr(self._o, {"pyr_down_u16": "pyr_down"}.get(name, name))


INTEGER_OUTPUTS = {"pyr_down_u16", "bgr_to_intensity", "sobel_dx", "sobel_dy", "pyr_down_uchar_gauss", "residual_sums", "residual_valid",
                   "residual_zero", "residual_one"}


def bits_differ(a, b):
    """number of elements whose bit patterns differ (NaN == NaN when both are NaN)"""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype.kind == "f":
        na, nb = np.isnan(a), np.isnan(b)
        u = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
        return int(((a.view(u) != b.view(u)) & ~(na & nb)).sum())
    return int((a != b).sum())


def max_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    m = ~(np.isnan(a) & np.isnan(b))
    if not m.any():
        return 0.0
    scale = np.maximum(np.abs(b[m]).max(), 1e-30)
    return float(np.abs(a[m] - b[m]).max() / scale)
