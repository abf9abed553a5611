"""SURVEY §8f row 4, the optimiser: the product's built-in solver of the local deformation graph (ef_solve_local_deformation, a
from-scratch banded Gauss-Newton; elasticfusion_amd/csrc/ef_deform_solver.hpp) against the reference's own — Core/Deformation.cpp,
Core/Utils/DeformationGraph.cpp and CholeskyDecomp.cpp compiled where they lie (oracle/Makefile `refsolver`; CHOLMOD replaced by a dense
stand-in, the graph nodes fed through the tape-recorder GL).  Same nodes, same constraints, same time stamps: the two graphs
(rotation, translation of every node, as floats) must agree to 1e-6; on well-conditioned inputs they are identical."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libefr_solver.so")
P = C.c_void_p


def have():
    if not os.path.exists(SO) and os.path.isdir("/root/reference/Core"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refsolver"])
    return os.path.exists(SO)


pytestmark = pytest.mark.skipif(not have(), reason="oracle/_ref/libefr_solver.so can only be built where /root/reference exists")


def reference(nodes, cons, time, prior=0):
    so = C.CDLL(SO)
    so.efs_local_constrain.argtypes = [P, C.c_int, P, C.c_int, C.c_int, C.c_int, P, P]
    g = np.zeros((1024, 16), np.float32)
    k = C.c_int(0)
    ok = so.efs_local_constrain(nodes.ctypes.data, len(nodes), cons.ctypes.data, len(cons), time, prior, g.ctypes.data, C.byref(k))
    return (g[:k.value].copy() if ok else None)


def problem(seed, n, m, pin, spread=0.05, shift=(0.004, -0.003, 0.002)):
    rng = np.random.RandomState(seed)
    s = np.linspace(0, 1, n)
    nodes = np.zeros((n, 4), np.float32)
    nodes[:, 0] = 2 * s - 1 + rng.normal(0, 0.01, n)
    nodes[:, 1] = 0.3 * np.sin(4 * s) + rng.normal(0, 0.01, n)
    nodes[:, 2] = 1.5 + 0.2 * np.cos(3 * s)
    nodes[:, 3] = np.cumsum(rng.randint(1, 9, n))                    # strictly ascending creation times
    pick = rng.randint(0, n, m)
    cons = np.zeros((m, 8))
    cons[:, 0:3] = nodes[pick, :3] + rng.normal(0, spread, (m, 3))
    a = 0.01
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    cons[:, 3:6] = cons[:, 0:3] @ Rz.T + np.asarray(shift) + rng.normal(0, 0.0003, (m, 3))
    cons[:, 6] = np.maximum(1, np.floor(nodes[pick, 3] * 0.3))        # the inactive surface is older
    cons[:, 7] = pin
    return nodes, cons, int(nodes[-1, 3]) + 3


@pytest.mark.parametrize("case", [dict(seed=1, n=40, m=60, pin=1), dict(seed=2, n=5, m=12, pin=1), dict(seed=3, n=200, m=300, pin=0),
                                  dict(seed=4, n=64, m=150, pin=1, spread=0.15), dict(seed=5, n=120, m=400, pin=0)])
def test_builtin_solver_matches_the_compiled_reference(case):
    from elasticfusion_amd import api
    nodes, cons, time = problem(**case)
    ref = reference(nodes, cons, time)
    got = api.solve_local_deformation(nodes, cons, time, 0)
    assert ref is not None and got is not None and len(ref) == len(got[0]) == len(nodes)
    g = got[0]
    assert np.array_equal(g[:, [0, 1, 2, 15]], ref[:, [0, 1, 2, 15]])
    assert np.abs(g[:, 3:15] - ref[:, 3:15]).max() <= 1e-6, np.abs(g[:, 3:15] - ref[:, 3:15]).max()
    moved = np.abs(ref[:, 12:15]).max()
    assert 5e-4 < moved < 5e-2                                        # the graph really deformed


def test_only_nodes_younger_than_the_last_deformation_move():
    from elasticfusion_amd import api
    nodes, cons, time = problem(seed=7, n=80, m=160, pin=0)
    prior = int(nodes[45, 3])
    ref = reference(nodes, cons, time, prior)
    g, _, _ = api.solve_local_deformation(nodes, cons, time, prior)
    assert np.abs(g[:, 3:15] - ref[:, 3:15]).max() <= 1e-6
    fixed = nodes[:, 3] <= prior
    ident = np.tile(np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float32), (int(fixed.sum()), 1))
    assert np.array_equal(g[fixed, 3:15], ident) and np.abs(g[~fixed, 12:15]).max() > 1e-4


def test_too_small_a_graph_is_refused():
    from elasticfusion_amd import api
    nodes, cons, time = problem(seed=8, n=4, m=10, pin=1)
    assert reference(nodes, cons, time) is None and api.solve_local_deformation(nodes, cons, time, 0) is None
