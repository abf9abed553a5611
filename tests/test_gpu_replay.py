"""GPU, front-end tier: a synthetic .klg (raw and zlib depth frames) replayed by the C++ headless front-end through
libefusion.so's class ElasticFusion must land on the same trajectory as the oracle run on the same frames."""
import os
import subprocess

import numpy as np
import pytest

import efo

pytestmark = pytest.mark.gpu


def test_klg_replay_through_cpp_shim(tmp_path, seq):
    from elasticfusion_amd import api, synth
    n = 6
    frames = [seq.frame(k) for k in range(n)]
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "efusion_replay")
    # like the reference's run loop the front-end never processes the LAST frame of a log (RawLogReader::hasMore)
    o = efo.Fusion()
    for k, (rgb, depth, _) in enumerate(frames[:-1]):
        o.process_frame(rgb, depth, k * 33333)
    Tr = o.pose()
    for compress in (False, True):
        log = str(tmp_path / f"seq{int(compress)}.klg")
        synth.write_klg(log, frames, compress_depth=compress)
        r = subprocess.run([exe, "-l", log, "-ply"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        words = r.stdout.split()
        assert int(words[1]) == n - 1 and int(words[words.index("surfels") + 1]) == o.map_count(), r.stdout
        traj = np.loadtxt(log + ".freiburg")
        assert traj.shape == (n - 1, 8)
        assert np.allclose(traj[:, 0], np.arange(n - 1) * 33333 / 1e6, atol=1e-6)
        assert np.abs(traj[-1, 1:4] - Tr[:3, 3]).max() <= 1e-6     # six significant digits, as the reference writes them
        # savePly keeps surfels above the confidence threshold only (ElasticFusion.cpp:703-712): a 6-frame map has none
        # yet, the header must still be a valid binary PLY
        hdr = open(log + ".ply", "rb").read(400)
        assert hdr.startswith(b"ply\nformat binary_little_endian 1.0") and b"element vertex" in hdr and b"end_header" in hdr


def test_klg_replay_with_close_loops(tmp_path, seq):
    """-cl: the C++ class constructed with closeLoops = true is the reference's closed-loop mode — fern database (keyframes stored at the
    end of every frame, matched mid-frame), global closure, local closure with the built-in optimiser — and equals the oracle's frame
    loop in the same configuration (fern table from the class's fixed seed 0, the same optimiser behind the oracle's solver hook)."""
    from elasticfusion_amd import api, synth
    from test_gpu_global import oracle_with_ferns
    n = 8
    frames = [seq.frame(k) for k in range(n)]
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "efusion_replay")
    o, calls = oracle_with_ferns(0, timeDelta=3, confidence=2.0)
    attempts = opened = 0
    for k, (rgb, depth, _) in enumerate(frames):
        o.process_frame(rgb, depth, k * 33333)
        info, _ = o.local_loop()
        attempts += info.attempted
        opened += info.gates_ok
    log = str(tmp_path / "loops.klg")
    synth.write_klg(log, frames)
    # (the front-end's own defaults are MainController's -ic 40000 -ie 4e-05; the oracle above runs on the constructor's)
    r = subprocess.run([exe, "-l", log, "-cl", "-t", "3", "-c", "2", "-all", "-ic", "35000", "-ie", "5e-05"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("local loop closure")][0].split()
    assert int(line[line.index("attempts") + 1]) == attempts == n - 1 and int(line[line.index("open") + 1]) == opened
    fern_line = [ln for ln in r.stdout.splitlines() if ln.startswith("fern database")][0].split()
    assert int(fern_line[fern_line.index("keyframes") + 1]) == len(o.ferns()) >= 1
    words = r.stdout.split()
    assert int(words[words.index("surfels") + 1]) == o.map_count(), r.stdout
    traj = np.loadtxt(log + ".freiburg")
    assert np.abs(traj[-1, 1:4] - o.pose()[:3, 3]).max() <= 1e-6
    # getLocalDeformation().getGraph() as MainController draws it (:388-404): Deformation::sampleGraphModel's nodes of the final map, four
    # sequence neighbours each
    import efo
    nodes = efo.sample_graph(o.map())
    g = [ln for ln in r.stdout.splitlines() if ln.startswith("deformation graph")][0].split()
    assert int(g[g.index("nodes") + 1]) == len(nodes) > 4 and int(g[g.index("links") + 1]) == 4 * len(nodes), (g, len(nodes))
    first = [float(x) for x in g[g.index("first") + 1:g.index("first") + 4]]
    assert np.abs(np.asarray(first) - nodes[0, :3]).max() <= 1e-6 * max(1.0, float(np.abs(nodes[0, :3]).max())), (first, nodes[0])


def test_jpeg_klg_replay(tmp_path, seq):
    """BASELINE.json configs[1] is a replay of a recorded log, whose colour frames are JPEG images: the front-end decodes them through
    the system libjpeg (include/efusion_jpeg.hpp; the decoder is pinned against the reference's own reader in tests/test_klg_vs_reference.py)
    and the run equals the oracle's on the same decoded frames."""
    pytest.importorskip("PIL")
    import ctypes as C
    from elasticfusion_amd import api, synth
    n = 5
    frames = [seq.frame(k) for k in range(n)]
    log = str(tmp_path / "jpeg.klg")
    synth.write_klg(log, frames, compress_depth=True, jpeg_quality=92)
    so = C.CDLL(os.path.join(os.path.dirname(api.LIB_PATH), "libefusion.so"))
    so.efk_open.restype = C.c_void_p
    h = C.c_void_p(so.efk_open(log.encode(), 640, 480, 0, 0))
    o = efo.Fusion()
    k = 0
    while so.efk_has_more(h):
        ts = C.c_int64(0)
        depth = np.zeros((480, 640), np.uint16)
        rgb = np.zeros((480, 640, 3), np.uint8)
        assert so.efk_next(h, C.byref(ts), depth.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p)) == 1
        assert np.array_equal(depth, frames[k][1]) and not np.array_equal(rgb, frames[k][0])
        o.process_frame(rgb, depth, ts.value)
        k += 1
    so.efk_close(h)
    assert k == n - 1
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "efusion_replay")
    r = subprocess.run([exe, "-l", log], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    words = r.stdout.split()
    assert int(words[1]) == n - 1 and int(words[words.index("surfels") + 1]) == o.map_count(), r.stdout
    traj = np.loadtxt(log + ".freiburg")
    assert np.abs(traj[-1, 1:4] - o.pose()[:3, 3]).max() <= 1e-6
