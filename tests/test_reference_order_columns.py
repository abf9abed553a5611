"""CPU: the work split of the reference-order persistent tracker (k_track_ref, elasticfusion_amd/csrc/ef_track_ref_persistent.inc), with the
product's own mapping functions (rt_column / rt_vwarp / rt_pixel: cut out of the source and compiled for the host):

  * workgroup -> column is a bijection on 0..255, XCD-contiguous (workgroup w runs on XCD w % 8: its column lies in [32 (w % 8), 32 (w % 8) + 32));
  * column 4 b + m owns virtual warps m and m + 4 of reference block b — the pair the first level of blockReduceSum's 8-warp tree adds
    (reduce.cu:97-117) — so the 256 columns cover the 512 virtual warps exactly once;
  * the visits idx < 64 K of a column (K = ceil(N / 16384) passes) are pass idx / 64 of its 64 virtual threads: over all columns every pixel
    of an N-pixel level is visited exactly once — what the correspondence search relies on (a workgroup writes exactly the packed
    correspondences its own photometric wavefronts read) and what makes the sums the reference's (virtual thread g owns pixels g, g + 16384, ...).
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "elasticfusion_amd", "csrc")


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("rtcols"))
    src = open(os.path.join(CSRC, "ef_track_ref_persistent.inc")).read()
    a, b = src.index("__device__ __forceinline__ int rt_column("), src.index("// warpReduceSum over each 32-lane half of the wavefront")   # (the wave-level sums behind it are DPP moves: device only)
    cut = src[a:b]
    assert "rt_vwarp" in cut and "rt_pixel" in cut
    wgs = int(re.search(r"constexpr int FT_WGS = (\d+)", open(os.path.join(CSRC, "ef_track_exchange.inc")).read()).group(1))
    vth = int(re.search(r"constexpr int VTHREADS = (\d+)", open(os.path.join(CSRC, "ef_track.hpp")).read()).group(1))
    host = os.path.join(tmp, "rt.cpp")
    open(host, "w").write("#define __device__\n#define __forceinline__ inline\nconstexpr int FT_WGS = %d, VTHREADS = %d;\n%s\n" % (wgs, vth, cut) +
                          'extern "C" int col_of(int wg) { return rt_column(wg); }\nextern "C" int warp_of(int col, int wl) { return rt_vwarp(col, wl); }\n'
                          'extern "C" int pixel_of(int col, int idx) { return rt_pixel(col, idx); }\n')
    so = os.path.join(tmp, "rt.so")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", host, "-o", so], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    return C.CDLL(so), wgs, vth


def test_columns_are_an_xcd_contiguous_bijection_over_the_pairs_of_the_reference_blocks(lib):
    L, wgs, vth = lib
    cols = [L.col_of(w) for w in range(wgs)]
    assert sorted(cols) == list(range(wgs))
    for w, c in enumerate(cols):
        assert 32 * (w % 8) <= c < 32 * (w % 8) + 32, (w, c)
    warps = []
    for c in range(wgs):
        w0, w1 = L.warp_of(c, 0), L.warp_of(c, 1)
        assert w0 // 8 == w1 // 8 == c // 4 and w0 % 8 == c % 4 and w1 == w0 + 4, (c, w0, w1)     # warps m and m + 4 of block c / 4
        warps += [w0, w1]
    assert sorted(warps) == list(range(vth // 32))


@pytest.mark.parametrize("n", [640 * 480, 320 * 240, 160 * 120, 80 * 60, 20 * 15, 1280 * 960, 332 * 252, 100 * 76, 1, 16384, 16385])
def test_every_pixel_of_a_level_is_visited_exactly_once(lib, n):
    L, wgs, vth = lib
    K = (n + vth - 1) // vth
    seen = np.zeros(n, np.int32)
    for c in range(wgs):
        for idx in range(64 * K):
            p = L.pixel_of(c, idx)
            assert p % vth == L.warp_of(c, (idx >> 5) & 1) * 32 + (idx & 31) and p // vth == idx >> 6   # pass idx / 64 of virtual thread idx % 64
            if p < n:
                seen[p] += 1
    assert (seen == 1).all(), (n, int((seen != 1).sum()))
