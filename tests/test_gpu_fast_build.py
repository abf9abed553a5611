"""GPU: the OPT-IN fast build (libefusion_hip_fast.so: fused multiply-adds + the fast summation order + the whole tracker as one persistent
launch) against ITS specification (libefo_oracle_fast.so), bit for bit.

Since round 5 the default pair of every other test file is the reference rounding (libefusion_hip.so == libefo_oracle.so == the
compiled reference); this file re-runs, under the fast pair, the tests that hold the fast build's own machinery: the operator-tier sums
(k_se3_accum_fast, fast_tree), the frame tier (k_track_fast, persistent vs launch-per-step), the tracker's configuration space, the
one-workgroup fallback when the chip is partly taken (tests/test_gpu_fallback.py), a closed-loop run (the model-to-model tracker's empty-view shortcut) and the
130-frame steady state.  Its tie to the REFERENCE is tests/test_oracle_vs_reference.py (the fast oracle within 3e-5 of the compiled
reference's sums, integer outputs identical) and the one-frame harness (tests/test_gpu_one_frame.py, profiles/r05_parity_factorial.json):
the fast build is NOT inside the 1e-4 m / 1e-4 rad bar on every frame, which is why it is opt-in."""
import numpy as np
import pytest

import efo

pytestmark = [pytest.mark.gpu, pytest.mark.fastbuild]


def test_operator_tier_sums_equal_the_fast_oracle(fast_pair, frames):
    import test_gpu_ops_tracking as T
    f = efo.Fusion()
    for k in range(3):
        rgb, depth, _ = frames[k]
        f.process_frame(rgb, depth, k)
    odo = f.odometry()
    T.test_icp_step(fast_pair.ops, odo, f)
    T.test_rgb_residual_and_step(fast_pair.ops, odo)
    T.test_so3_step(fast_pair.ops, odo)


def test_frames_equal_the_fast_oracle(fast_pair, seq):
    import test_gpu_frame as T
    T.test_tracking_and_fusion_sequence(fast_pair, seq)


@pytest.mark.parametrize("cfg", [dict(), dict(icpThresh=100.0), dict(so3=False), dict(fastOdom=True)], ids=["default", "icp_only", "no_so3", "fastOdom"])
def test_persistent_launch_and_launch_per_step_agree(fast_pair, seq, cfg):
    import test_gpu_frame as T
    T.test_persistent_and_per_step_tracker_scripts_agree(fast_pair, seq, cfg)


@pytest.mark.parametrize("name", ["fastOdom", "icp_only", "no_pyramid", "rgb_only", "low_confidence_time_window"])
def test_tracker_configurations_equal_the_fast_oracle(fast_pair, seq, name):
    import test_gpu_frame as T
    assert name in T.VARIANTS, sorted(T.VARIANTS)
    T.test_tracker_configurations_match_oracle(fast_pair, seq, name)


def test_graph_replay_and_two_contexts(fast_pair, seq):
    import test_gpu_frame as T
    T.test_graph_replayed_tracker_matches_oracle(fast_pair, seq)
    T.test_two_contexts_interleaved(fast_pair)


def test_closed_loop_front_half_equals_the_fast_oracle(fast_pair):
    import test_gpu_loop as T
    opened, applied, sh, so = T.run_pair(__import__("loopscene").OneShotSolver)
    assert opened >= 2 and applied == 1


def test_steady_state_equals_the_fast_oracle(fast_pair):
    import test_gpu_steady as T
    frames = T.make_sequences(("clean",))["clean"]
    h = T.run_hip(fast_pair, frames)
    o = T.run_oracle(frames)
    T.assert_same_run(h, o, "fast")
