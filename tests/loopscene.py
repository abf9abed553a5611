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
