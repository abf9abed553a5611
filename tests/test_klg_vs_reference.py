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


@pytest.mark.parametrize("flip", [False, True])
def test_jpeg_colour_frames_match_the_compiled_reference(libs, tmp_path, flip):
    """JPEG-compressed colour (RawLogReader.cpp:94-96 -> JPEGLoader.h:40-90): both readers hand the same stream to the same libjpeg
    runtime, so the bytes must be identical — including the R/B exchange JPEGLoader applies to what the decoder returns (quirk Q5) —
    and an independent decoder (Pillow's) confirms which way round that is."""
    pytest.importorskip("PIL")
    import io
    from PIL import Image
    from elasticfusion_amd import synth
    mine, ref = libs
    seq = synth.Sequence(seed=0xEF0006, width=W, height=H)
    frames = [seq.frame(k) for k in range(4)]
    log = str(tmp_path / "j.klg")
    synth.write_klg(log, frames, compress_depth=True, jpeg_quality=90)
    n_ref, got_ref = read_all(ref, "efrk_", C.c_void_p(ref.efrk_open(log.encode(), W, H, int(flip))))
    h = mine.efk_open(log.encode(), W, H, 0, int(flip))
    assert h, mine.efk_last_error()
    n, got = read_all(mine, "efk_", C.c_void_p(h))
    assert n == n_ref == 4 and len(got) == len(got_ref) == 3
    for k, ((ta, da, ca), (tb, db, cb)) in enumerate(zip(got, got_ref)):
        assert ta == tb and np.array_equal(da, db) and np.array_equal(da, frames[k][1])
        assert np.array_equal(ca, cb), k
        buf = io.BytesIO()
        Image.fromarray(frames[k][0], "RGB").save(buf, format="JPEG", quality=90)
        pil = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
        want = pil if flip else pil[..., ::-1]        # JPEGLoader stores {t2, t1, t0}; flipColors swaps back
        assert np.abs(ca.astype(int) - want.astype(int)).max() <= 2, k     # two libjpeg builds may round the IDCT differently by one
        src = frames[k][0].astype(int)                # lossy, but unmistakably the source with (without) its channels exchanged
        d_swapped, d_plain = np.abs(ca.astype(int) - src[..., ::-1]).mean(), np.abs(ca.astype(int) - src).mean()
        assert (d_plain < d_swapped) if flip else (d_swapped < d_plain), (d_swapped, d_plain)


def test_reader_errors(libs, tmp_path):
    mine, _ = libs
    assert mine.efk_open(str(tmp_path / "missing.klg").encode(), W, H, 0, 0) is None
    assert b"cannot open" in mine.efk_last_error()
