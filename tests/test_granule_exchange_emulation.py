"""The persistent tracker's granule exchanges, executed WITHOUT a GPU (tests/wave_emu/exchange_host.cpp): the product's own task plan and column
mapping (fast_plan, fast_groups_per_xcd, fast_group_of: ef_track_fast.inc; ft_column: ef_track_fast_persistent.inc — cut out of the sources)
under a workgroup-level restatement of exchanges A and B, one host thread per workgroup.

What is pinned:
  * the column map is a bijection at every level (groups first, in group order: the tree's order; pixel-less workgroups behind them);
  * with EVERY workgroup publishing into every exchange (the product), single-buffered slots + a fresh epoch per exchange + `tag == epoch`
    waits are safe under arbitrary skew — also across level changes (NG 75 -> 150 -> 240 -> 253) and with one workgroup asleep for many
    iterations' worth of time: nobody times out, every total is right;
  * the protocol the first version of the kernel had — pixel-less workgroups only LISTEN — is not: a listener that falls behind is lapped
    (nobody depends on it, so nobody waits for it) and waits for an epoch that has been overwritten.  The GPU showed it as time-outs with
    two interleaved contexts (tests/test_gpu_one_frame.py::test_checkpoint_resume_continues_the_replay_bit_for_bit); here it is deterministic."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "elasticfusion_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "wave_emu")
LEVELS = [160 * 120, 320 * 240, 640 * 480, 1280 * 960]      # NG = 75, 150, 240, 253


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("exchange"))
    a = open(os.path.join(CSRC, "ef_track_fast.inc")).read()
    b = open(os.path.join(CSRC, "ef_track_fast_persistent.inc")).read()
    i, j = a.index("struct FastPlan {"), a.index("// lane 0 receives the adjacent-pair tree")
    k, l = b.index("__device__ __forceinline__ int ft_column("), b.index("// the reducer's part of exchange B")
    text = a[i:j] + b[k:l]
    assert "fast_group_of" in text and "ft_column" in text and "fast_plan" in text
    x = open(os.path.join(CSRC, "ef_track_exchange.inc")).read()   # the exchanges themselves (shared by both summation orders since round 5)
    wgs = int(re.search(r"constexpr int FT_WGS = (\d+)", x).group(1))
    cut = os.path.join(tmp, "exchange_cut.inc")
    open(cut, "w").write(text)
    so = os.path.join(tmp, "exchange.so")
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-DFT_WGS=%d" % wgs, '-DEXCHANGE_SOURCE="%s"' % cut, os.path.join(EMU, "exchange_host.cpp"), "-o", so]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    L = C.CDLL(so)
    L.run_exchange.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    L.exchange_columns.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return L, wgs


@pytest.mark.parametrize("n", LEVELS + [80 * 60, 100 * 76, 332 * 252, 64, 1 << 22])
def test_column_map_is_a_bijection_with_the_groups_first(lib, n):
    L, wgs = lib
    cols, groups, ng = (C.c_int * wgs)(), (C.c_int * wgs)(), C.c_int(0)
    L.exchange_columns(n, cols, groups, C.byref(ng))
    cols, groups = np.array(cols), np.array(groups)
    assert sorted(cols) == list(range(wgs))
    owners = groups >= 0
    assert owners.sum() == ng.value and sorted(groups[owners]) == list(range(ng.value))
    assert np.array_equal(cols[owners], groups[owners]) and (cols[~owners] >= ng.value).all()
    # XCD-contiguous: the workgroups of XCD-slot x (w % 8 == x) own a contiguous range of groups
    for x in range(8):
        g = np.sort(groups[(np.arange(wgs) % 8 == x) & owners])
        assert len(g) == 0 or np.array_equal(g, np.arange(g[0], g[0] + len(g)))


def run(L, iters, all_publish, spin, jitter_us=0, sleeper=-1, sleeper_at=0, sleep_ms=0, na=6):
    lv = (C.c_int * len(LEVELS))(*LEVELS)
    stats = (C.c_longlong * 3)()
    rc = L.run_exchange(iters, lv, len(LEVELS), na, int(all_publish), spin, jitter_us, sleeper, sleeper_at, sleep_ms, stats)
    return rc, list(stats)


def test_every_workgroup_publishing_is_safe_under_skew_and_level_changes(lib):
    L, wgs = lib
    rc, s = run(L, iters=60, all_publish=True, spin=1 << 24)
    assert rc == 0 and s == [60, 0, 0], s
    rc, s = run(L, iters=40, all_publish=True, spin=1 << 24, jitter_us=200)
    assert rc == 0 and s == [40, 0, 0], s
    # workgroup 255 (pixel-less at every level but the last) sleeps 100 ms before iteration 5: everybody waits for it, nothing breaks
    rc, s = run(L, iters=24, all_publish=True, spin=1 << 24, sleeper=wgs - 1, sleeper_at=5, sleep_ms=100)
    assert rc == 0 and s == [24, 0, 0], s


def test_listen_only_workgroups_are_lapped(lib):
    L, wgs = lib
    # the same sleeper under the first version's protocol: at iteration 4 (level 160x120: 75 groups) workgroup 255 has no pixels, nobody waits
    # for it, the others run on and overwrite the epochs it still waits for
    rc, s = run(L, iters=24, all_publish=False, spin=1 << 14, sleeper=wgs - 1, sleeper_at=4, sleep_ms=300)
    assert rc == 1 and s[1] >= 1, s
