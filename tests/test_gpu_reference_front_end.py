"""GPU: the REFERENCE's own front end executing frames through this library.

oracle/_ref/reference_front_end is the reference's Main.cpp + MainController.cpp + Tools/RawLogReader.cpp + Core/Utils/Parse.cpp, compiled
where they lie against include/ElasticFusion.h and linked with libefusion.so (`make -C oracle reffrontend`; the binary travels to the GPU
box, /root/reference does not).  Its run loop (MainController.cpp:222-254: logReader->getNext(), eFusion->processFrame(...), the
destructor's .freiburg) constructs `new ElasticFusion(...)` with MainController's own arguments (:178-194) and replays a synthetic .klg:

* open loop (-o) and closed loop (the reference's default: fern database + global + local closure);
* the .freiburg it leaves must be byte-identical to the one tools/efusion_replay.cpp leaves with the same settings, and equal the
  oracle's trajectory pose by pose to the six significant digits the reference prints.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

import efo

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_EXE = os.path.join(ROOT, "oracle", "_ref", "reference_front_end")
N = 12


def _run(cmd, cwd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=cwd, timeout=240)
    assert r.returncode == 0, (cmd, r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    return r


@pytest.fixture(scope="module")
def logs(tmp_path_factory, seq):
    from elasticfusion_amd import synth
    if not os.path.exists(REF_EXE):
        pytest.fail("oracle/_ref/reference_front_end is missing: run `make -C oracle reffrontend` where /root/reference exists "
                    "(__graft_entry__.build() does); the binary travels to the GPU box with the snapshot")
    d = tmp_path_factory.mktemp("rfe")
    frames = [seq.frame(k) for k in range(N)]
    a, b = str(d / "a.klg"), str(d / "b.klg")
    synth.write_klg(a, frames)
    shutil.copy(a, b)
    return str(d), a, b, frames


def _same_as_oracle(traj, o_poses):
    assert traj.shape == (len(o_poses), 8)
    assert np.allclose(traj[:, 0], np.arange(len(o_poses)) * 33333 / 1e6, atol=1e-6)
    for k, T in enumerate(o_poses):
        # the reference prints six significant digits (operator<< of a stringstream, ElasticFusion.cpp:141-155)
        assert np.abs(traj[k, 1:4] - T[:3, 3]).max() <= 2e-6 * max(1.0, float(np.abs(T[:3, 3]).max())), (k, traj[k], T[:3, 3])


def test_reference_front_end_replays_open_loop(logs):
    from elasticfusion_amd import api
    d, a, b, frames = logs
    replay = os.path.join(os.path.dirname(api.LIB_PATH), "efusion_replay")
    _run([REF_EXE, "-l", a, "-q", "-o"], d)
    _run([replay, "-l", b, "-q", "-o"], d)
    fa, fb = open(a + ".freiburg", "rb").read(), open(b + ".freiburg", "rb").read()
    assert len(fa) > 0 and fa == fb, "the reference's front end and efusion_replay left different trajectories"
    # like RawLogReader::hasMore the run loop never delivers the last frame of a log
    o = efo.Fusion()
    poses = []
    for k, (rgb, depth, _) in enumerate(frames[:-1]):
        o.process_frame(rgb, depth, k * 33333)
        poses.append(o.pose().copy())
    _same_as_oracle(np.loadtxt(a + ".freiburg"), poses)
    os.rename(a + ".freiburg", a + ".open.freiburg")
    os.remove(b + ".freiburg")


def test_reference_front_end_replays_closed_loop(logs):
    """No -o: MainController constructs the class with closeLoops = true, its own -ic 40000 / -ie 4e-05 thresholds, timeDelta 200,
    confidence 10 (MainController.cpp:60-110,178-194)."""
    from elasticfusion_amd import api
    from test_gpu_global import oracle_with_ferns
    d, a, b, frames = logs
    replay = os.path.join(os.path.dirname(api.LIB_PATH), "efusion_replay")
    _run([REF_EXE, "-l", a, "-q"], d)
    _run([replay, "-l", b, "-q", "-cl"], d)
    fa, fb = open(a + ".freiburg", "rb").read(), open(b + ".freiburg", "rb").read()
    assert len(fa) > 0 and fa == fb, "closed loop: the reference's front end and efusion_replay left different trajectories"
    o, _calls = oracle_with_ferns(0)
    o.set_close_loops(True, icpCountThresh=40000, icpErrThresh=4e-05)
    poses = []
    for k, (rgb, depth, _) in enumerate(frames[:-1]):
        o.process_frame(rgb, depth, k * 33333)
        poses.append(o.pose().copy())
    _same_as_oracle(np.loadtxt(a + ".freiburg"), poses)
