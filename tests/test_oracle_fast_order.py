"""CPU: THE FAST ORDER (the summation order of the opt-in fast build libefusion_hip_fast.so, oracle/efo_track.cpp built -DEFO_FAST_ORDER = libefo_oracle_fast.so) against an independent numpy restatement of its
specification, bit for bit: row-groups of 64 pixels; U = max(1, ceil(RG / 1024)) row-groups per task; leaf (task, lane) adds its pixels in
order; the total is the complete adjacent-pair binary tree over the leaves in index order task * 64 + lane (missing leaves = +0).
The HIP kernels are held to the oracle's sums bit for bit (tests/test_gpu_ops_tracking.py, test_gpu_frame.py, ...); this file holds the
oracle to the text."""
import ctypes as C

import numpy as np
import pytest

import efo


def spec_sum(v):
    v = np.asarray(v, np.float32)
    n = len(v)
    rg = (n + 63) // 64
    u = max(1, (rg + 1023) // 1024)
    t = (rg + u - 1) // u
    pad = np.zeros(t * u * 64, np.float32)
    pad[:n] = v
    a = pad.reshape(t, u, 64)
    leaves = np.zeros((t, 64), np.float32)
    for k in range(u):                       # lane-wise, in visit order (float32 adds)
        leaves = (leaves + a[:, k, :]).astype(np.float32)
    s = leaves.reshape(-1)
    while len(s) > 1:
        if len(s) % 2:
            s = np.concatenate([s, np.zeros(1, np.float32)])
        s = (s[0::2] + s[1::2]).astype(np.float32)
    return s[0]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4800, 7600, 19200, 76800, 100 * 76, 307200, 332 * 252, 1228800])
def test_fast_order_sum_equals_the_specification(n):
    import subprocess
    subprocess.check_call(["make", "-s", "-C", efo.ORACLE_DIR, "libefo_oracle_fast.so"])
    lib = C.CDLL(efo.FAST_SO)
    lib.efo_fast_order_sum.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.efo_fast_order_sum.restype = None
    rng = np.random.RandomState(n)
    v = (rng.standard_normal(n) * np.exp(rng.uniform(-6, 6, n))).astype(np.float32)   # wide dynamic range: every association shows
    out = np.zeros(1, np.float32)
    lib.efo_fast_order_sum(v.ctypes.data, n, out.ctypes.data)
    want = spec_sum(v)
    assert out[0].view(np.uint32) == np.float32(want).view(np.uint32), (n, out[0], want)
    # ... and it IS an order: the plain left-to-right sum differs somewhere in this family
    if n == 307200:
        assert np.float32(np.cumsum(v, dtype=np.float32)[-1]) != out[0]


def test_the_task_plan_of_the_levels_the_engine_runs():
    """(U, tasks, groups of four tasks) as ef_track_fast.inc::fast_plan has them: 640x480 -> 5 / 960 / 240 at level 0"""
    def plan(n):
        rg = (n + 63) // 64
        u = max(1, (rg + 1023) // 1024)
        t = (rg + u - 1) // u
        return u, t, (t + 3) // 4
    assert plan(640 * 480) == (5, 960, 240) and plan(320 * 240) == (2, 600, 150) and plan(160 * 120) == (1, 300, 75)
    assert plan(1280 * 960) == (19, 1011, 253) and plan(80 * 60) == (1, 75, 19)
    assert all(plan(n)[2] <= 256 for n in range(1, 1 << 22, 4099))     # never more groups than the persistent launch has workgroups
