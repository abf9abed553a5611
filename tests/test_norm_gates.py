"""The ICP gates of the two-visits-per-lane rows (round 6; elasticfusion_amd/csrc/ef_track_kernels.hip, icp2_stage2b) compare SQUARED norms with
thresholds computed once per call, where icp_row (reduce.cu:258-266) compares the norms themselves: `dist <= distThres`, `sine < angleThres`.
sqrtf is correctly rounded and monotone, so a largest float a with sqrtf(a) <= T (< T) exists and the two forms agree on EVERY input — this
test runs the product's own sq_le_max / sq_lt_max (ef_device.hpp, compiled for the host) against the square-root form around the boundary of
the reference's two thresholds, of edge-case thresholds and of 200 000 random ones (subnormal, huge, infinite, NaN)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "elasticfusion_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "wave_emu")


def test_squared_norm_gates_equal_the_square_root_gates(tmp_path):
    so = str(tmp_path / "norm_gates.so")
    cmd = ["g++", "-std=c++20", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I" + EMU, "-I" + CSRC, os.path.join(EMU, "norm_gates_host.cpp"), "-o", so]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    lib = C.CDLL(so)
    lib.norm_gates_check.restype = C.c_long
    n, le, lt = C.c_long(0), C.c_float(0), C.c_float(0)
    bad = lib.norm_gates_check(200000, C.byref(n), C.byref(le), C.byref(lt))
    assert bad == 0 and n.value > 2.5e7
    # the reference's thresholds (RGBDOdometry.h:41-42): 0.10 m and sin(20 deg)
    assert le.value.hex() == "0x1.47ae160000000p-7" and lt.value.hex() == "0x1.df240a0000000p-4"
