"""Synthetic RGB-D sequences with analytic geometry and scripted SE(3) motion.

Scene: interior of a 4 x 2.5 x 3 m box with three spheres, closed-form ray casting => exact depth.
Depth is ``round(1000 z)`` uint16 mm, kept only inside the accepted 300..3000 mm band (0 = invalid,
depth_bilateral.frag:34 / MainController.cpp:70).  Colour is a world-anchored procedural albedo
(sinusoid mixture + value noise, period ~8-40 px) with no zero bytes (0 means "invalid" to the
tracker, reduce.cu:647,679).  Trajectory: smooth Lissajous, <= 8 mm and <= 0.4 deg per frame.

This module only *generates inputs* (it is what a .klg replay would feed processFrame); it is not a
checker and does not touch oracle/.
"""
from __future__ import annotations

import numpy as np

DEFAULT_INTRINSICS = dict(width=640, height=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0)  # MainController.cpp:37-43


def _rot_xyz(rx: float, ry: float, rz: float) -> np.ndarray:
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


class Sequence:
    """Frames k = 0.. of one synthetic sequence.  ``seed`` 0xEF0001..0xEF0008 select the 8 bench sequences."""

    def __init__(self, seed: int = 0xEF0001, width: int = 640, height: int = 480, fx: float | None = None,
                 fy: float | None = None, cx: float | None = None, cy: float | None = None, noise: bool = False,
                 speed: float = 1.0):
        s = width / 640.0
        self.width, self.height = width, height
        self.fx = 528.0 * s if fx is None else fx
        self.fy = 528.0 * s if fy is None else fy
        self.cx = 320.0 * s if cx is None else cx
        self.cy = 240.0 * s if cy is None else cy
        self.noise = noise
        self.speed = speed
        rng = np.random.RandomState(seed & 0x7FFFFFFF)
        self._phase = rng.uniform(0, 2 * np.pi, size=6)
        self._amp_t = np.array([0.25, 0.10, 0.18]) * rng.uniform(0.8, 1.0, size=3)
        self._amp_r = np.array([0.12, 0.30, 0.08]) * rng.uniform(0.8, 1.0, size=3)
        self._per = np.array([400.0, 300.0, 500.0, 350.0, 450.0, 600.0]) * rng.uniform(0.9, 1.1, size=6)
        self._noise_rng = np.random.RandomState(0xEF5107)
        # box half extents (x, y, z) and spheres (centre, radius), world frame = camera frame at k = 0
        self.box = np.array([2.0, 1.25, 1.5])
        self.spheres = [(np.array([-0.7, 0.45, 1.0]), 0.35), (np.array([0.6, 0.6, 0.9]), 0.30),
                        (np.array([0.1, -0.35, 1.15]), 0.25)]
        v, u = np.mgrid[0:height, 0:width]
        self._dirs = np.stack([(u - self.cx) / self.fx, (v - self.cy) / self.fy, np.ones_like(u, dtype=np.float64)], -1)
        self._T0_inv = np.linalg.inv(self._abs_pose(0))
        # value-noise lattice
        self._lattice = rng.uniform(-1, 1, size=(32, 32, 32, 3))

    # -- trajectory -------------------------------------------------------------------------------
    def _abs_pose(self, k: int) -> np.ndarray:
        kk = k * self.speed
        a = 2 * np.pi * kk / self._per + self._phase
        t = self._amp_t * np.sin(a[:3])
        r = self._amp_r * np.sin(a[3:])
        T = np.eye(4)
        T[:3, :3] = _rot_xyz(*r)
        T[:3, 3] = t
        return T

    def pose(self, k: int) -> np.ndarray:
        """Ground-truth T_wc of frame k with the world frame = camera frame of frame 0 (4x4 float64)."""
        return self._T0_inv @ self._abs_pose(k)

    # -- rendering --------------------------------------------------------------------------------
    def _albedo(self, p: np.ndarray) -> np.ndarray:
        x, y, z = p[..., 0], p[..., 1], p[..., 2]
        out = np.empty(p.shape[:-1] + (3,), dtype=np.float64)
        # angular frequencies in rad/m: spatial periods ~3-12 cm == ~10-40 px at 1.5 m (1 px ~ 2.8 mm)
        freqs = [((55.1, 23.7, 31.3), (97.3, -61.1, 44.9), (-46.7, 123.1, 72.3)),
                 ((47.7, 58.9, -34.1), (-83.9, 45.1, 109.7), (135.3, 47.3, -59.1)),
                 ((35.9, -46.1, 59.7), (111.1, 84.3, -56.3), (-58.3, -97.9, 85.1))]
        # trilinear value noise on a 4 cm lattice
        q = p / 0.04
        q0 = np.floor(q).astype(np.int64)
        f = q - q0
        f = f * f * (3 - 2 * f)
        L = self._lattice

        def lat(dx, dy, dz):
            return L[(q0[..., 0] + dx) % 32, (q0[..., 1] + dy) % 32, (q0[..., 2] + dz) % 32]

        fx, fy, fz = f[..., 0:1], f[..., 1:2], f[..., 2:3]
        n = (((lat(0, 0, 0) * (1 - fx) + lat(1, 0, 0) * fx) * (1 - fy) + (lat(0, 1, 0) * (1 - fx) + lat(1, 1, 0) * fx) * fy) * (1 - fz)
             + ((lat(0, 0, 1) * (1 - fx) + lat(1, 0, 1) * fx) * (1 - fy) + (lat(0, 1, 1) * (1 - fx) + lat(1, 1, 1) * fx) * fy) * fz)
        for c in range(3):
            acc = 0.0
            for (a, b, d), wgt in zip(freqs[c], (38.0, 26.0, 18.0)):
                acc = acc + wgt * np.sin(a * x + b * y + d * z + c)
            out[..., c] = 128.0 + acc + 30.0 * n[..., c]
        return np.clip(np.rint(out), 1, 255).astype(np.uint8)

    def frame(self, k: int):
        """Returns (rgb uint8 [H,W,3], depth uint16 [H,W] in mm, T_wc 4x4) for frame k."""
        T = self._abs_pose(k)
        R, o = T[:3, :3], T[:3, 3]
        d = self._dirs @ R.T  # world ray directions; parameter t == camera-space z
        # box (we are inside): per axis the exit plane
        with np.errstate(divide="ignore", invalid="ignore"):
            tb = np.where(d > 0, (self.box - o) / d, (-self.box - o) / d)
        tb = np.where(np.isfinite(tb), tb, np.inf)
        t = tb.min(-1)
        for c, r in self.spheres:
            oc = o - c
            a = (d * d).sum(-1)
            b = 2 * (d @ oc)
            cc = oc @ oc - r * r
            disc = b * b - 4 * a * cc
            ok = disc > 0
            sq = np.sqrt(np.where(ok, disc, 0))
            ts = (-b - sq) / (2 * a)
            ts = np.where(ok & (ts > 1e-6), ts, np.inf)
            t = np.minimum(t, ts)
        p = o + d * t[..., None]
        rgb = self._albedo(p)
        z = t.copy()
        if self.noise:
            z = z + self._noise_rng.normal(0, 1.0, size=z.shape) * 0.0012 * z * z
        mm = np.rint(1000.0 * z)
        depth = np.where((mm >= 300) & (mm <= 3000), mm, 0).astype(np.uint16)
        return rgb, depth, self.pose(k)


class ClutterSequence(Sequence):
    """Second scene family (round 4): planar clutter + thin structures + a sensor-like depth channel.

    Same room, trajectory family and albedo as ``Sequence``, without the spheres; instead
      * PLANAR CLUTTER: a dozen finite rectangular boards standing and lying in the room at 0.8-2.6 m, arbitrary orientation
        (large coplanar regions, many depth discontinuities, boards partly occluding one another);
      * THIN STRUCTURES: poles and bars of 1-2.5 cm radius (a few pixels wide: surfels that live on 2-6 px of support);
      * the depth a structured-light sensor would report: quantised in DISPARITY (1/8 px of a 580 px * 75 mm rig: 3 mm steps at 1 m,
        3 cm at 3 m), invalid (0) on depth discontinuities and at grazing incidence, rectangular drop-outs that move from frame to
        frame, and — with ``noise`` — the same z^2 range noise as ``Sequence``.
    Every frame draws its randomness from (seed, k): frame(k) does not depend on which frames were rendered before."""

    def __init__(self, seed: int = 0xEF0001, **kw):
        super().__init__(seed, **kw)
        rng = np.random.RandomState((seed ^ 0xC1A77E5) & 0x7FFFFFFF)
        self.spheres = []
        self.boards = []      # (centre, unit normal, unit u, unit v, half u, half v)
        for _ in range(12):
            c = np.array([rng.uniform(-0.85, 0.85), rng.uniform(-0.6, 0.6), rng.uniform(0.65, 1.3)])
            n = rng.normal(size=3) * np.array([0.5, 0.4, 1.0])
            n /= np.linalg.norm(n)
            u = np.cross(n, rng.normal(size=3))
            u /= np.linalg.norm(u)
            v = np.cross(n, u)
            self.boards.append((c, n, u, v, rng.uniform(0.10, 0.30), rng.uniform(0.08, 0.25)))
        self.rods = []        # (point, unit axis, radius, half length)
        for _ in range(9):
            c = np.array([rng.uniform(-0.9, 0.9), rng.uniform(-0.6, 0.6), rng.uniform(0.6, 1.35)])
            a = np.array([0.0, 1.0, 0.0]) if rng.rand() < 0.6 else rng.normal(size=3)
            a = a / np.linalg.norm(a)
            self.rods.append((c, a, rng.uniform(0.010, 0.025), rng.uniform(0.4, 1.2)))
        self._seed = seed

    def frame(self, k: int):
        T = self._abs_pose(k)
        R, o = T[:3, :3], T[:3, 3]
        d = self._dirs @ R.T
        with np.errstate(divide="ignore", invalid="ignore"):
            tb = np.where(d > 0, (self.box - o) / d, (-self.box - o) / d)
        tb = np.where(np.isfinite(tb), tb, np.inf)
        t = tb.min(-1)
        axis = tb.argmin(-1)
        nrm = np.zeros(d.shape)
        np.put_along_axis(nrm, axis[..., None], -np.sign(np.take_along_axis(d, axis[..., None], -1)), -1)
        with np.errstate(divide="ignore", invalid="ignore"):
            for c, n, u, v, hu, hv in self.boards:
                den = d @ n
                tt = ((c - o) @ n) / den
                p = o + d * tt[..., None] - c
                hit = (np.abs(den) > 1e-9) & (tt > 1e-6) & (np.abs(p @ u) <= hu) & (np.abs(p @ v) <= hv) & (tt < t)
                t = np.where(hit, tt, t)
                nrm = np.where(hit[..., None], n * -np.sign(den)[..., None], nrm)
            for c, a, r, hl in self.rods:
                oc = o - c
                dp = d - (d @ a)[..., None] * a
                ocp = oc - (oc @ a) * a
                A = (dp * dp).sum(-1)
                B = 2 * (dp @ ocp)
                Cc = ocp @ ocp - r * r
                disc = B * B - 4 * A * Cc
                ok = (disc > 0) & (A > 1e-12)
                tt = (-B - np.sqrt(np.where(ok, disc, 0))) / (2 * A)
                p = o + d * tt[..., None] - c
                hit = ok & (tt > 1e-6) & (np.abs(p @ a) <= hl) & (tt < t)
                t = np.where(hit, tt, t)
                pn = p - (p @ a)[..., None] * a
                nrm = np.where(hit[..., None], pn / np.maximum(np.linalg.norm(pn, axis=-1, keepdims=True), 1e-12), nrm)
        p = o + d * t[..., None]
        rgb = self._albedo(p)
        z = t.copy()
        rng = np.random.RandomState(((self._seed * 2654435761) ^ (k * 40503 + 17)) & 0x7FFFFFFF)
        if self.noise:
            z = z + rng.normal(0, 1.0, size=z.shape) * 0.0012 * z * z
        # disparity quantisation (1/8 px, f * b = 580 px * 0.075 m)
        fb = 580.0 * 0.075
        z = fb / (np.rint(fb / z * 8.0) / 8.0)
        mm = np.rint(1000.0 * z)
        valid = (mm >= 300) & (mm <= 3000)
        # no return on depth discontinuities (> 4 cm to a 4-neighbour) and at grazing incidence (> 80 degrees)
        jump = np.zeros(z.shape, bool)
        jump[:, 1:] |= np.abs(t[:, 1:] - t[:, :-1]) > 0.04
        jump[:, :-1] |= np.abs(t[:, 1:] - t[:, :-1]) > 0.04
        jump[1:, :] |= np.abs(t[1:, :] - t[:-1, :]) > 0.04
        jump[:-1, :] |= np.abs(t[1:, :] - t[:-1, :]) > 0.04
        cosi = np.abs((d * nrm).sum(-1)) / np.linalg.norm(d, axis=-1)
        valid &= ~jump & (cosi > np.cos(np.deg2rad(80.0)))
        for _ in range(6):   # drop-outs
            h, w = z.shape
            y0, x0 = rng.randint(0, h - 8), rng.randint(0, w - 8)
            valid[y0:y0 + rng.randint(4, max(5, h // 10)), x0:x0 + rng.randint(4, max(5, w // 10))] = False
        depth = np.where(valid, mm, 0).astype(np.uint16)
        return rgb, depth, self.pose(k)


def make_sequence(seed: int = 0xEF0001, scene: str = "box", **kw) -> "Sequence":
    """scene 'box' = Sequence (box + spheres, exact depth), 'clutter' = ClutterSequence"""
    return ClutterSequence(seed, **kw) if scene == "clutter" else Sequence(seed, **kw)


def sample_surfels(seq: "Sequence", n: int = 1 << 20, radius: float = 0.004, conf: float = 12.0, init_time: int = 1, last_time: int = 1) -> np.ndarray:
    """~n surfels [m, 12] float32 sampled on the scene's surfaces ("ef_map_upload of surfels sampled on the scene
    surfaces, radius 4 mm, conf 12, times in window"), in ``seq``'s world frame (= its camera frame at k = 0), laid out as the map holds
    them: {x, y, z, confidence} {colour, 0, initTime, lastTime} {nx, ny, nz, radius}.  Regular grids on the six walls (area-proportional) and
    on the three spheres, wall after wall, row after row: neighbouring surfels are neighbours in memory, as in a map that grew frame by
    frame.  Normals point away from the room's interior, the way normals computed from a depth image point away from the camera."""
    bx, by, bz = seq.box
    walls = []   # (origin, u axis, v axis, outward normal), absolute scene frame
    for sgn in (-1.0, 1.0):
        walls.append((np.array([sgn * bx, -by, -bz]), np.array([0, 2 * by, 0.0]), np.array([0, 0, 2 * bz]), np.array([sgn, 0, 0.0])))
        walls.append((np.array([-bx, sgn * by, -bz]), np.array([2 * bx, 0, 0.0]), np.array([0, 0, 2 * bz]), np.array([0, sgn, 0.0])))
        walls.append((np.array([-bx, -by, sgn * bz]), np.array([2 * bx, 0, 0.0]), np.array([0, 2 * by, 0.0]), np.array([0, 0, sgn])))
    areas = [np.linalg.norm(u) * np.linalg.norm(v) for _, u, v, _ in walls] + [4 * np.pi * r * r for _, r in seq.spheres]
    pitch = np.sqrt(sum(areas) / n)
    P, Nn = [], []
    for o, u, v, nrm in walls:
        nu, nv = max(1, int(round(np.linalg.norm(u) / pitch))), max(1, int(round(np.linalg.norm(v) / pitch)))
        a, b = np.meshgrid((np.arange(nu) + 0.5) / nu, (np.arange(nv) + 0.5) / nv, indexing="ij")
        P.append(o + a[..., None] * u + b[..., None] * v)
        Nn.append(np.broadcast_to(nrm, P[-1].shape))
    for c, r in seq.spheres:
        m = max(8, int(round(4 * np.pi * r * r / (pitch * pitch))))
        i = np.arange(m) + 0.5
        phi, th = np.arccos(1 - 2 * i / m), np.pi * (1 + 5 ** 0.5) * i
        d = np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], -1)
        P.append(c + r * d)
        Nn.append(-d)   # the camera is outside the sphere: away from it = into the sphere
    P = np.concatenate([x.reshape(-1, 3) for x in P])
    Nn = np.concatenate([np.asarray(x, np.float64).reshape(-1, 3) for x in Nn])
    rgb = seq._albedo(P).astype(np.int64)
    R0, t0 = seq._T0_inv[:3, :3], seq._T0_inv[:3, 3]
    out = np.zeros((len(P), 12), np.float32)
    out[:, :3] = P @ R0.T + t0
    out[:, 3] = conf
    out[:, 4] = ((rgb[:, 0] << 16) + (rgb[:, 1] << 8) + rgb[:, 2]).astype(np.float32)   # color.glsl:19-34
    out[:, 6], out[:, 7] = init_time, last_time
    out[:, 8:11] = Nn @ R0.T
    out[:, 11] = radius
    return out


def write_klg(path: str, frames, timestamps=None, compress_depth: bool = False, jpeg_quality: int | None = None) -> None:
    """Writes frames [(rgb HxWx3 u8, depth HxW u16, ...)] as a .klg log (layout of Tools/RawLogReader.cpp:29,63-109):
    int32 numFrames, then per frame int64 timestamp, int32 depthSize, int32 imageSize, depth bytes (raw little-endian
    u16, or one zlib stream when ``compress_depth``), image bytes (raw RGB8: imageSize == 3*W*H means "not JPEG"; with
    ``jpeg_quality`` one baseline JPEG image per frame, as the reference's recorder writes them — needs Pillow)."""
    import io
    import struct
    import zlib
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(frames)))
        for k, fr in enumerate(frames):
            rgb, depth = np.ascontiguousarray(fr[0], np.uint8), np.ascontiguousarray(fr[1], "<u2")
            d = depth.tobytes()
            if compress_depth:
                d = zlib.compress(d)
            ts = int(timestamps[k]) if timestamps is not None else k * 33333
            img = rgb.tobytes()
            if jpeg_quality is not None:
                from PIL import Image
                buf = io.BytesIO()
                Image.fromarray(rgb, "RGB").save(buf, format="JPEG", quality=int(jpeg_quality))
                img = buf.getvalue()
                assert len(img) != rgb.nbytes
            f.write(struct.pack("<qii", ts, len(d), len(img)))
            f.write(d)
            f.write(img)
