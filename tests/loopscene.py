"""A synthetic revisit for the local loop closure (ElasticFusion.cpp:447-527): the camera maps view A, looks away at view B for
longer than the time window, and comes back to A with a few millimetres of drift.  The surfels of A that B never saw are then
INACTIVE; re-observing A lays new ACTIVE surfels over them, and the model-to-model registration between the two recovers the
drift.  Poses are injected (in_T_wc) so that both implementations see exactly the same camera path."""
import numpy as np

TIME_DELTA = 4
CONFIDENCE = 2.0
KS = list(range(0, 5)) + list(range(200, 207)) + list(range(5, 13))
DRIFT_FROM = 12


def drift(T):
    d = np.eye(4)
    a = 0.003
    d[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    d[:3, 3] = [0.004, -0.003, 0.002]
    return T @ d


def frames(width=640, height=480):
    from elasticfusion_amd import synth
    seq = synth.Sequence(seed=0xEF0001, width=width, height=height)
    out = []
    for i, k in enumerate(KS):
        rgb, depth, T = seq.frame(k)
        out.append((rgb, depth, drift(T) if i >= DRIFT_FROM else T))
    return out


def graph_from_constraints(cons, n_nodes=16):
    """Stand-in for Deformation::constrain: nodes on a subset of the constraint sources, each translating its neighbourhood
    onto the constraint's target; layout of GlobalModel.cpp:540-546 {position 3, rotation 9 column-major, translation 3, time}."""
    pick = np.linspace(0, len(cons) - 1, n_nodes).astype(int)
    g = np.zeros((n_nodes, 16), np.float32)
    g[:, 0:3] = cons[pick, 0:3]
    g[:, 3] = g[:, 7] = g[:, 11] = 1.0
    g[:, 12:15] = cons[pick, 3:6] - cons[pick, 0:3]
    g[:, 15] = np.sort(cons[pick, 6])
    return g


class OneShotSolver:
    """accepts the first attempt with at least `min_constraints` constraints, rejects every later one"""

    def __init__(self, min_constraints=100):
        self.min_constraints = min_constraints
        self.calls = []
        self.accepted = 0

    def __call__(self, info, cons):
        self.calls.append((info.n_constraints, cons.copy()))
        if self.accepted or len(cons) < self.min_constraints:
            return None
        self.accepted += 1
        return graph_from_constraints(cons)


class SweepSequence:
    """Closed-loop steady state at the reference's DEFAULT thresholds (tests/test_gpu_closed_steady.py): dwell on view A until its surfels are
    stable (confidence 10 takes some forty frames at a quarter of the pixels per frame), sweep away by 30 degrees, dwell on view B for longer
    than the time window — A's surfels outside B go INACTIVE by themselves —, sweep back and dwell: the inactive surface is re-observed, the
    model-to-model registration finds tens of thousands of inliers and the reference's own gates (35000 / 5e-5 / 1e-5) open.  No injected
    pose, no setTick, no relaxed gate.  (A wrapper, not a subclass, so that worker processes can rebuild it from its arguments.)"""
    PLAN = (90, 30, 30, 30, 14)     # dwell A, sweep, dwell B, sweep back, dwell A
    A = 0.26                        # half sweep, rad
    TIME_DELTA = 25

    def __init__(self, seed=0xEF0003):
        from elasticfusion_amd import synth
        self.seq = synth.Sequence(seed)
        self.seq._abs_pose = self._abs_pose
        self.seq._T0_inv = np.linalg.inv(self._abs_pose(0))
        self.n = sum(self.PLAN)

    def _abs_pose(self, k):
        from elasticfusion_amd import synth
        d0, r1, d1, r2, _ = self.PLAN
        if k < d0:
            th = -self.A
        elif k < d0 + r1:
            th = -self.A + 2 * self.A * (k - d0) / r1
        elif k < d0 + r1 + d1:
            th = self.A
        elif k < d0 + r1 + d1 + r2:
            th = self.A - 2 * self.A * (k - d0 - r1 - d1) / r2
        else:
            th = -self.A
        T = np.eye(4)
        T[:3, :3] = synth._rot_xyz(0.01 * np.sin(0.21 * k), th, 0.008 * np.sin(0.13 * k))
        T[:3, 3] = [0.03 * np.sin(0.17 * k), 0.02 * np.sin(0.11 * k), 0.02 * np.sin(0.07 * k)]
        return T

    def frame(self, k):
        return self.seq.frame(k)


def _sweep_frame(k):
    s = _sweep_frame.seq
    if s is None:
        s = _sweep_frame.seq = SweepSequence()
    return s.frame(k)


_sweep_frame.seq = None


def sweep_frames():
    """[(rgb, depth, T_wc)] of the whole sweep, rendered by spawned workers (the parent may already hold a HIP runtime)"""
    import multiprocessing as mp
    import os
    with mp.get_context("spawn").Pool(max(1, min(16, (os.cpu_count() or 2) - 1))) as pool:
        return pool.map(_sweep_frame, range(SweepSequence().n), chunksize=4)
