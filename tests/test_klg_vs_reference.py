"""the product's .klg reader (include/efusion_klg.hpp, exported from libefusion.so as efk_*) against the
reference's own Tools/RawLogReader.cpp compiled where it lies (oracle/Makefile `refklg`): the same frames, timestamps and bytes in
the same order for raw and zlib-compressed depth, with and without the R/B swap — including the reference's habit of never
delivering the last frame of a log (RawLogReader::hasMore is `currentFrame + 1 < numFrames`)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_KLG_SO = os.path.join(ROOT, "oracle", "_ref", "libefr_klg.so")
W, H = 160, 120


def have_reference_klg():
    if not os.path.exists(REF_KLG_SO) and os.path.isdir("/root/reference/Tools"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refklg"])
    return os.path.exists(REF_KLG_SO)


pytestmark = pytest.mark.skipif(not have_reference_klg(), reason="oracle/_ref/libefr_klg.so can only be built where /root/reference exists")


@pytest.fixture(scope="module")
def libs():
    from elasticfusion_amd import build
    build.build()
    mine = C.CDLL(os.path.join(ROOT, "elasticfusion_amd", "libefusion.so"))
    ref = C.CDLL(REF_KLG_SO)
    for so, pre in ((mine, "efk_"), (ref, "efrk_")):
        getattr(so, pre + "open").restype = C.c_void_p
        for f in ("close", "num_frames", "has_more", "next"):
            getattr(so, pre + f).argtypes = [C.c_void_p] + ([C.c_void_p] * 3 if f == "next" else [])
    mine.efk_last_error.restype = C.c_char_p
    return mine, ref


def read_all(so, pre, handle):
    out = []
    n = getattr(so, pre + "num_frames")(handle)
    while getattr(so, pre + "has_more")(handle):
        ts = C.c_int64(0)
        depth = np.zeros((H, W), np.uint16)
        rgb = np.zeros((H, W, 3), np.uint8)
        assert getattr(so, pre + "next")(handle, C.byref(ts), depth.ctypes.data, rgb.ctypes.data) == 1
        out.append((ts.value, depth, rgb))
    getattr(so, pre + "close")(handle)
    return n, out


@pytest.mark.parametrize("compress", [False, True])
@pytest.mark.parametrize("flip", [False, True])
def test_reader_matches_the_compiled_reference(libs, tmp_path, compress, flip):
    from elasticfusion_amd import synth
    mine, ref = libs
    seq = synth.Sequence(seed=0xEF0006, width=W, height=H)
    frames = [seq.frame(k) for k in range(5)]
    stamps = [1000 + 33333 * k for k in range(5)]
    log = str(tmp_path / "t.klg")
    synth.write_klg(log, frames, timestamps=stamps, compress_depth=compress)
    n_ref, got_ref = read_all(ref, "efrk_", C.c_void_p(ref.efrk_open(log.encode(), W, H, int(flip))))
    n, got = read_all(mine, "efk_", C.c_void_p(mine.efk_open(log.encode(), W, H, 0, int(flip))))
    assert n == n_ref == 5 and len(got) == len(got_ref) == 4          # the last frame is never delivered
    for k, ((ta, da, ca), (tb, db, cb)) in enumerate(zip(got, got_ref)):
        assert ta == tb == stamps[k]
        assert np.array_equal(da, db) and np.array_equal(da, frames[k][1])
        assert np.array_equal(ca, cb) and np.array_equal(ca, frames[k][0][..., ::-1] if flip else frames[k][0])
    # deliver_last_frame = 1: every frame
    n_all, got_all = read_all(mine, "efk_", C.c_void_p(mine.efk_open(log.encode(), W, H, 1, int(flip))))
    assert len(got_all) == 5 and got_all[-1][0] == stamps[-1] and np.array_equal(got_all[-1][1], frames[-1][1])


def test_reader_errors(libs, tmp_path):
    mine, _ = libs
    assert mine.efk_open(str(tmp_path / "missing.klg").encode(), W, H, 0, 0) is None
    assert b"cannot open" in mine.efk_last_error()
