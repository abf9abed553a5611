#!/usr/bin/env python
"""Generates tests/golden/deform_reference.npz from the REFERENCE's own graph optimiser compiled for the CPU (Core/Deformation.cpp +
Utils/DeformationGraph.cpp + Utils/CholeskyDecomp.cpp, oracle/_ref/libefr_solver.so, `make -C oracle refsolver`): four calls of
Deformation::constrain — a global closure with relative constraints (accepted), one below the entry gate and one inconsistent (both
rejected), and a local closure after an earlier deformation with keyframes — each with its inputs (nodes, constraint rows, poses) and
the reference's answers (accepted or not, the graph, the deformed poses, the relative constraints left behind).  Must be run where
/root/reference exists; tests/test_deform_golden.py replays them on ef_solve_deformation anywhere.

    python tools/make_deform_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_deform_global_vs_reference as T  # noqa: E402


def rows_array(rows):
    return np.array([list(s) + list(g) + [st, tt, rel, pin] for s, g, st, tt, rel, pin in rows], np.float64).reshape(-1, 10)


def main():
    g = {}
    nodes = T.path(200, 21)
    rng = np.random.RandomState(22)
    rel = []
    for a, b in [(150, 20), (151, 22), (120, 60), (121, 61), (90, 88), (40, 38)]:
        s = nodes[a, :3] + rng.normal(0, 0.03, 3)
        rel.append((s, s + rng.normal(0, 0.002, 3), int(nodes[a, 3]), int(nodes[b, 3]), 1, 0))
    fern = T.camera_poses(nodes, list(range(0, 200, 17)), 24)
    traj = T.camera_poses(nodes, list(range(0, 200, 2)), 25)
    cases = []
    rows, tick = T.closure(nodes, 23, 45, (0.12, 0.02, -0.06), 0.0, 0.0, relative_from=rel)
    cases.append((nodes, rows, tick, 1, 0, fern, traj))
    rows, tick = T.closure(nodes, 23, 45, (0.02, 0.0, 0.01), 0.0, 0.0)
    cases.append((nodes, rows, tick, 1, 0, fern, traj))
    small = T.path(60, 11)
    rows, tick = T.closure(small, 12, 30, (0.5, 0.3, -0.4), 0.4, 0.02)
    cases.append((small, rows, tick, 1, 0, T.camera_poses(small, [3, 30, 55], 1), T.camera_poses(small, list(range(0, 60, 4)), 2)))
    rows, tick = T.closure(nodes, 23, 45, (0.012, 0.002, -0.006), 0.0, 0.0, relative_from=rel)
    cases.append((nodes, [r for r in rows if not r[5]], tick, 0, int(nodes[100, 3]), fern, traj))
    for i, (nd, rows, tick, fm, last, fern, traj) in enumerate(cases):
        ok, graph, fp, tp, rel_out = T.reference(nd, rows, tick, fm, fm, last, fern[0], fern[1], traj[0], traj[1])
        p = f"c{i}_"
        g.update({p + "nodes": nd, p + "rows": rows_array(rows), p + "par": np.array([tick, fm, last, int(ok)]), p + "fern": fern[0], p + "fern_t": fern[1],
                  p + "traj": traj[0], p + "traj_t": traj[1], p + "graph": graph, p + "fern_out": fp, p + "traj_out": tp, p + "rel": rel_out})
        print(i, "accepted" if ok else "rejected", len(rows), "rows", len(rel_out), "relative left")
    path = os.path.join(ROOT, "tests", "golden", "deform_reference.npz")
    np.savez_compressed(path, n=np.array(len(cases)), **g)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
