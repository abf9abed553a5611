"""Known-answer properties of the built-in local-deformation optimiser (ef_solve_local_deformation; host code, runs without a GPU):
what it must do on inputs whose answer is known, independent of any reference build."""
import numpy as np
import pytest


def line_graph(n, seed=0):
    rng = np.random.RandomState(seed)
    s = np.linspace(0, 1, n)
    nodes = np.zeros((n, 4), np.float32)
    nodes[:, 0] = 3 * s - 1.5
    nodes[:, 1] = 0.2 * np.sin(5 * s) + rng.normal(0, 0.005, n)
    nodes[:, 2] = 1.2 + 0.3 * s
    nodes[:, 3] = 10 * np.arange(n) + 10      # (a node created at time 0 would not be younger than "no deformation yet" = 0)
    return nodes, rng


def apply_graph(g, p, carriers):
    """deform point p with nodes `carriers` = [(weight, index)] of graph g (rows: pos 3, R 9 column-major, t 3, time)"""
    out = np.zeros(3)
    for w, i in carriers:
        R = g[i, 3:12].reshape(3, 3).T
        out += w * (R @ (p - g[i, 0:3]) + g[i, 0:3] + g[i, 12:15])
    return out


def test_uniform_translation_is_recovered():
    from elasticfusion_amd import api
    nodes, rng = line_graph(60)
    shift = np.array([0.006, -0.004, 0.003])
    m = 240
    pick = rng.randint(0, 60, m)
    cons = np.zeros((m, 8))
    cons[:, 0:3] = nodes[pick, :3] + rng.normal(0, 0.03, (m, 3))
    cons[:, 3:6] = cons[:, 0:3] + shift
    cons[:, 6] = 1
    g, err, mce = api.solve_local_deformation(nodes, cons, int(nodes[-1, 3]) + 1, 0)
    covered = np.unique(pick)
    assert np.abs(g[covered, 12:15] - shift).max() < 5e-4                       # every constrained neighbourhood moved by the shift
    assert np.abs(g[:, 3:12] - np.tile(np.eye(3).T.reshape(9), (60, 1))).max() < 2e-3   # and did not rotate
    assert mce < 1e-4 and err < 1e-3


def test_constraints_end_up_closer_to_their_targets():
    from elasticfusion_amd import api
    nodes, rng = line_graph(120, seed=2)
    m = 300
    pick = rng.randint(0, 120, m)
    cons = np.zeros((m, 8))
    cons[:, 0:3] = nodes[pick, :3] + rng.normal(0, 0.04, (m, 3))
    a = 0.02
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    cons[:, 3:6] = cons[:, 0:3] @ Rz.T + [0.003, 0.001, -0.002]
    cons[:, 6] = 1
    before = np.linalg.norm(cons[:, 3:6] - cons[:, 0:3], axis=1).mean()
    g, err, mce = api.solve_local_deformation(nodes, cons, int(nodes[-1, 3]) + 1, 0)
    assert mce < 0.1 * before, (before, mce)
    # the graph stays close to rigid: columns of every node's matrix near-orthonormal
    for i in range(0, 120, 7):
        R = g[i, 3:12].reshape(3, 3).T
        assert np.abs(R.T @ R - np.eye(3)).max() < 5e-3


def test_nothing_to_optimise_returns_the_identity_graph():
    from elasticfusion_amd import api
    nodes, rng = line_graph(30)
    cons = np.zeros((10, 8))
    cons[:, 0:3] = nodes[:10, :3]
    cons[:, 3:6] = nodes[:10, :3] + 0.01
    cons[:, 6] = 1
    last = int(nodes[-1, 3])                                                     # every node is as old as the last deformation: none may move
    g, err, mce = api.solve_local_deformation(nodes, cons, last + 5, last)
    assert np.array_equal(g[:, 3:12], np.tile(np.eye(3, dtype=np.float32).reshape(9), (30, 1))) and not g[:, 12:15].any()
    assert np.array_equal(g[:, :3], nodes[:, :3]) and np.array_equal(g[:, 15], nodes[:, 3])


def test_refusals():
    from elasticfusion_amd import api
    nodes, _ = line_graph(30)
    cons = np.zeros((3, 8))
    assert api.solve_local_deformation(nodes[:4], cons, 1, 0) is None           # not more than k = 4 nodes: no graph (Deformation.cpp:283)
    assert api.solve_local_deformation(nodes, np.zeros((0, 8)), 1, 0) is None   # no constraints
