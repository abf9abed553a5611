"""Synthetic 80x60 views for the fern-database tests (tests/test_ferns_vs_reference.py, tests/test_ferns_golden.py,
tools/make_ferns_golden.py): `place(k)` is a smooth bumpy, textured surface distinct per k with a hole in its depth; `jitter`
re-draws a little sensor noise on top, which is what separates "the same place again" from "a new place"."""
import numpy as np

W, H = 640, 480
w, h = W // 8, H // 8
FX = FY = 528.0
CX, CY = 320.0, 240.0


def rot(axis, a):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def pose(axis, a, t):
    T = np.eye(4)
    T[:3, :3] = rot(axis, a)
    T[:3, 3] = t
    return T


def place(k, jitter=0, holes=True):
    """one 80x60 view: a smooth bumpy surface with a smooth texture, distinct per place k; `jitter` re-draws small noise"""
    rng = np.random.default_rng(1000 + k)
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    ph = rng.uniform(0, 6.28, 8)
    fr = rng.uniform(0.05, 0.25, 8)
    z = 1.6 + 0.5 * np.sin(fr[0] * u + ph[0]) * np.cos(fr[1] * v + ph[1]) + 0.4 * np.sin(fr[2] * (u + v) + ph[2]) + 0.1 * k % 0.7
    rgb = np.stack([127 + 120 * np.sin(fr[3 + c] * u + ph[3 + c]) * np.cos(fr[(5 + c) % 8] * v + ph[(5 + c) % 8]) for c in range(3)], -1)
    if jitter:
        jr = np.random.default_rng(77 * k + jitter)
        z = z + jr.normal(0, 0.002, z.shape)
        rgb = rgb + jr.normal(0, 2.0, rgb.shape)
    z = z.astype(np.float32)
    if holes:
        z[(u - 20 - 3 * (k % 5)) ** 2 + (v - 25) ** 2 < 36] = 0
    return (np.clip(rgb, 0, 255).astype(np.uint8),) + geometry(z)


def geometry(z):
    """vertex and normal images (f32x4) of a depth image at the 1/8-resolution intrinsics"""
    z = np.ascontiguousarray(z, np.float32)
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    verts = np.zeros((h, w, 4), np.float32)
    verts[..., 0] = ((u - CX / 8) / (FX / 8)).astype(np.float32) * z
    verts[..., 1] = ((v - CY / 8) / (FY / 8)).astype(np.float32) * z
    verts[..., 2] = z
    verts[..., 3] = (z > 0)
    norms = np.zeros((h, w, 4), np.float32)
    norms[..., 2] = -1
    return verts, norms
