"""Driver of tests/test_host_logic_sanitized.py::test_klg_reader_survives_mutated_logs — run as a script under LD_PRELOAD=libasan:libubsan
with the sanitized build of the .klg reader's C API as argv[1].  Writes three small logs (raw, zlib depth, zlib depth + JPEG colour),
replays each, then replays MUTATED copies: truncated anywhere, bytes flipped in the first frame's header, bytes flipped anywhere, a 32-bit
field overwritten with 0x7fffffff / 0xffffffff (frame counts and block sizes the reader must not trust).  The reader may refuse a file
or stop early; it must not read or write out of bounds, overflow, or hang — the sanitizers abort the process if it does."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elasticfusion_amd import synth  # noqa: E402

W, H, FRAMES = 64, 48, 4


def main(so, trials):
    rng = np.random.default_rng(1)
    frames = [(rng.integers(0, 255, (H, W, 3), dtype=np.uint8), rng.integers(0, 4000, (H, W), dtype=np.uint16), k * 33333) for k in range(FRAMES)]
    lib = C.CDLL(so)
    lib.efk_open.restype = C.c_void_p
    lib.efk_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.efk_close.argtypes = [C.c_void_p]
    lib.efk_num_frames.argtypes = [C.c_void_p]
    lib.efk_has_more.argtypes = [C.c_void_p]
    lib.efk_next.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]

    def replay(path):
        r = lib.efk_open(path.encode(), W, H, 1, 0)
        if not r:
            return -1, 0
        n = lib.efk_num_frames(r)
        ts = C.c_int64()
        d = np.zeros((H, W), np.uint16)
        c = np.zeros((H, W, 3), np.uint8)
        got = 0
        for _ in range(max(0, min(n, 64)) + 2):
            if not lib.efk_has_more(r) or not lib.efk_next(r, C.byref(ts), d.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)):
                break
            got += 1
        lib.efk_close(r)
        return n, got

    tmp = tempfile.mkdtemp(prefix="klg_fuzz_")
    base = {}
    for name, kw in (("raw", {}), ("zlib", dict(compress_depth=True)), ("jpeg", dict(compress_depth=True, jpeg_quality=90))):
        p = os.path.join(tmp, name + ".klg")
        synth.write_klg(p, frames, **kw)
        base[name] = open(p, "rb").read()
        assert replay(p) == (FRAMES, FRAMES), (name, replay(p))
    refused = short = full = 0
    for name, data in base.items():
        for trial in range(trials):
            b = bytearray(data)
            mode = trial % 4
            if mode == 0:
                b = b[:rng.integers(0, len(b))]
            elif mode == 1:
                for _ in range(rng.integers(1, 6)):
                    b[rng.integers(0, min(len(b), 200))] = rng.integers(0, 256)
            elif mode == 2:
                for _ in range(rng.integers(1, 20)):
                    b[rng.integers(0, len(b))] = rng.integers(0, 256)
            else:
                off = int(rng.integers(0, max(1, len(b) - 4)))
                b[off:off + 4] = (0xFFFFFFFF if trial % 8 == 3 else 0x7FFFFFFF).to_bytes(4, "little")
            p = os.path.join(tmp, "mutated.klg")
            open(p, "wb").write(bytes(b))
            n, got = replay(p)
            refused += n < 0
            short += n >= 0 and got < FRAMES
            full += got >= FRAMES
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    print("KLG_FUZZ_OK refused %d short %d full %d" % (refused, short, full))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
