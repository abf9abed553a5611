"""The product's HOST logic under AddressSanitizer + UndefinedBehaviorSanitizer (gcc's runtimes), without a GPU.

libefusion_hip.so is built by hipcc, whose sanitizer runtimes are not in this image; but the parts of it that never touch the device are
plain C++ and compile with g++ as they stand:
  * elasticfusion_amd/csrc/ef_ferns.hip (the fern database, the closure object: keyframes, trajectory, relative constraints, the decisions
    of a global / local closure and of relocalisation) with ef_deform_solver.hpp (the deformation-graph optimiser) and the solver's C entry
    points cut out of ef_context.hip;
  * include/efusion_klg.hpp (the .klg reader, which parses files from outside) with its C API cut out of efusion_shim.hip.
Two checks, each in a child process with the sanitizer runtimes preloaded:
  1. a subset of the CPU parity tests of that logic (golden sessions of the compiled reference, the closure decisions against the oracle)
     runs against the sanitized build (EF_HIP_LIB, the harness's library override) — same answers, no report;
  2. the .klg reader replays mutated logs (tests/klg_fuzz.py): it may refuse or stop early, it must not touch memory it does not own.
Any report (heap / stack overflow, use after free, signed overflow, misaligned or null access, out-of-range shift or cast) fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "elasticfusion_amd", "csrc")


def runtimes():
    libs = []
    for name in ("libasan.so", "libubsan.so"):
        p = subprocess.run(["gcc", "-print-file-name=" + name], stdout=subprocess.PIPE, text=True).stdout.strip()
        if not os.path.isabs(p) or not os.path.exists(p):
            return None
        libs.append(p)
    return ":".join(libs)


pytestmark = pytest.mark.skipif(runtimes() is None, reason="gcc's sanitizer runtimes are not installed")
SAN = ["-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off"]


def child_env(**extra):
    env = dict(os.environ, LD_PRELOAD=runtimes(), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:allocator_may_return_null=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0")
    env.update(extra)
    return env


def assert_clean(r):
    out = r.stdout
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-6000:]
    assert r.returncode == 0, out[-6000:]


def test_fern_database_closure_and_graph_optimiser_under_sanitizers(tmp_path):
    hip = tmp_path / "hip"
    hip.mkdir()
    # ef_linalg_dev.hpp includes <hip/hip_runtime.h> for the execution-space keywords only
    (hip / "hip_runtime.h").write_text("#pragma once\n#define __host__\n#define __device__\n#define __forceinline__ inline\n#define __global__\n")
    ctx = open(os.path.join(CSRC, "ef_context.hip")).read()
    a, b = ctx.index("int ef_solve_local_deformation(const float* nodes4"), ctx.index("int ef_enable_global_closure(ef_ctx* c")
    assert "ef_solve_deformation_gated" in ctx[a:b] and "hip" not in ctx[a:b].lower().replace("ef_hip", "")
    (tmp_path / "solver_entry_points.cpp").write_text(
        '#include <cstdint>\n#include <vector>\n#include "%s"\n#include "%s"\nextern "C" {\n%s}\n'
        'extern "C" const char* ef_last_error(const ef_ctx*) { return ""; }\nextern "C" void* ef_stream(ef_ctx*) { return nullptr; }\n'
        % (os.path.join(ROOT, "include", "ef_hip.h"), os.path.join(CSRC, "ef_deform_solver.hpp"), ctx[a:b]))
    so = str(tmp_path / "libhost_logic_san.so")
    r = subprocess.run(["g++", "-x", "c++", *SAN, "-w", "-I" + str(tmp_path), os.path.join(CSRC, "ef_ferns.hip"), str(tmp_path / "solver_entry_points.cpp"), "-o", so],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    tests = ["tests/test_ferns_golden.py", "tests/test_ferns_coded.py", "tests/test_deform_golden.py", "tests/test_deform_solver.py",
             "tests/test_closure_vs_oracle.py::test_global_closure_decisions"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-s", "-x", "-p", "no:cacheprovider", *tests], cwd=ROOT, env=child_env(EF_HIP_LIB=so),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert_clean(r)
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:]


def test_klg_reader_survives_mutated_logs(tmp_path):
    shim = open(os.path.join(CSRC, "efusion_shim.hip")).read()
    api = shim[shim.index("// ---- C API of the .klg reader"):]
    assert "efk_open" in api and "efk_next" in api
    (tmp_path / "klg_api.cpp").write_text('#include <cstring>\n#include <stdexcept>\n#include <string>\n#include "%s"\n%s'
                                          % (os.path.join(ROOT, "include", "efusion_klg.hpp"), api))
    so = str(tmp_path / "libklg_san.so")
    r = subprocess.run(["g++", *SAN, str(tmp_path / "klg_api.cpp"), "-o", so, "-lz", "-ldl"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "klg_fuzz.py"), so, "240"], cwd=ROOT, env=child_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    assert_clean(r)
    assert "KLG_FUZZ_OK" in r.stdout, r.stdout[-3000:]
