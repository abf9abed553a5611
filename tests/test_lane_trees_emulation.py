"""The fast order's wavefront trees, executed WITHOUT a GPU: lane_tree64 (one accumulator, six cross-lane steps) and lane_trees (all
accumulators of a wavefront at once: the lanes of a pair split them between each other, level by level — a third of the instructions) are
cut out of the product's sources and run by 64 host threads, one per lane (tests/wave_emu/lane_trees_host.cpp emulates the DPP / permlane /
swizzle builtins).  Both must form, for every accumulator, the adjacent-pair binary tree over the 64 lanes the specification asks for
(tests/test_oracle_fast_order.py), bit for bit — on values with a wide dynamic range, where every association shows."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "elasticfusion_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "wave_emu")


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("lane_trees"))
    k = open(os.path.join(CSRC, "ef_track_kernels.hip")).read()
    f = open(os.path.join(CSRC, "ef_track_fast.inc")).read()
    a, b = k.index("__device__ __forceinline__ float quad_xor1("), k.index("template <int Q>\n__device__ __forceinline__ float quad_bcast(")
    c, d = k.index("__device__ __forceinline__ float down32("), k.index("// 4x4 transpose across (lane-in-quad, register)")
    e, g = f.index("// lane 0 receives the adjacent-pair tree over the 64 lanes"), f.index("// JtJJtrSE3::add (types.cuh:104-143")
    text = k[a:b] + k[c:d] + f[e:g]
    assert all(n in text for n in ("quad_xor1", "quad_xor2", "down32", "down16", "row_down", "lane_tree64", "lane_split_step", "lane_trees_store"))
    cut = os.path.join(tmp, "lane_trees_cut.inc")
    open(cut, "w").write(text)
    so = os.path.join(tmp, "lane_trees.so")
    cmd = ["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", '-DLANE_TREES_SOURCE="%s"' % cut,
           os.path.join(EMU, "lane_trees_host.cpp"), "-o", so]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    L = C.CDLL(so)
    L.run_lane_trees.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return L


def pair_tree(col):
    s = np.asarray(col, np.float32)
    while len(s) > 1:
        s = (s[0::2] + s[1::2]).astype(np.float32)
    return s[0]


@pytest.mark.parametrize("na", [29, 11])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_lane_trees_form_the_adjacent_pair_tree(lib, na, seed):
    rng = np.random.RandomState(seed * 100 + na)
    v = (rng.standard_normal((64, na)) * np.exp(rng.uniform(-8, 8, (64, na)))).astype(np.float32)
    if seed == 3:
        v[rng.rand(64, na) < 0.5] = 0.0          # rejected pixels: exact zeros in the chains
    t64, tt = np.full(na, -1, np.float32), np.full(na, -1, np.float32)
    lib.run_lane_trees(v.ctypes.data, na, t64.ctypes.data, tt.ctypes.data)
    want = np.array([pair_tree(v[:, a]) for a in range(na)], np.float32)
    assert np.array_equal(t64.view(np.uint32), want.view(np.uint32)), (t64, want)
    assert np.array_equal(tt.view(np.uint32), want.view(np.uint32)), (tt, want)
    assert not np.array_equal(want, v.astype(np.float64).sum(0).astype(np.float32)) or seed == 3   # (an order, not just a sum)
