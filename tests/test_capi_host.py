"""CPU-only checks of the drop-in boundary: libefusion_hip.so loads, exports every symbol that include/ef_hip.h
declares, refuses to run without a GPU (no silent fallback), and the host-side mirror never routes through oracle/."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    from elasticfusion_amd import api, build
    if not os.path.exists(api.LIB_PATH):
        build.build()
    return C.CDLL(api.LIB_PATH)


def test_header_symbols_are_exported(so):
    hdr = open(os.path.join(ROOT, "include", "ef_hip.h")).read()
    names = sorted(set(re.findall(r"\b(ef_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) > 60
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing


def test_default_config_matches_front_end_defaults(so):
    from elasticfusion_amd import api
    cfg = api.ef_config()
    so.ef_default_config(C.byref(cfg))
    assert (cfg.width, cfg.height) == (640, 480)                      # MainController.cpp:37
    assert (cfg.fx, cfg.fy, cfg.cx, cfg.cy) == (528.0, 528.0, 320.0, 240.0)  # MainController.cpp:42
    assert cfg.confidence == 10.0 and cfg.depth_cut == 3.0 and cfg.icp_weight == 10.0
    assert cfg.time_delta == 2147483647 // 2 and cfg.so3 == 1 and cfg.pyramid == 1 and cfg.close_loops == 0


def test_create_fails_loudly_without_gpu_or_with_bad_config(so):
    from elasticfusion_amd import api
    so.ef_last_error.restype = C.c_char_p
    so.ef_last_error.argtypes = [C.c_void_p]
    h = C.c_void_p()
    cfg = api.default_config(width=641)
    assert so.ef_create(C.byref(cfg), C.byref(h)) == -1
    if not os.path.exists("/dev/kfd"):
        cfg = api.default_config()
        rc = so.ef_create(C.byref(cfg), C.byref(h))
        assert rc == -2 and b"HIP" in so.ef_last_error(None)  # EF_EHIP: no device => error, never a CPU path
        with pytest.raises(api.EFError):
            api.ElasticFusion()


def test_product_sources_never_reference_the_oracle():
    bad = []
    for base in ("elasticfusion_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle/|libefo_oracle|import efo|efo_[a-z]+\(", txt) and f not in ("__init__.py",):
                        if "oracle/" in txt and "does not touch oracle/" in txt and len(re.findall(r"oracle/", txt)) == 1:
                            continue
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_cpp_shim_library_and_replay_tool_exist_and_fail_loudly(tmp_path):
    """libefusion.so (class ElasticFusion of include/ElasticFusion.h) and the headless replay front-end are built by
    build(); without a GPU the front-end must exit with an error, not fall back."""
    import subprocess
    import numpy as np
    from elasticfusion_amd import api, build, synth
    build.build()
    shim = os.path.join(os.path.dirname(api.LIB_PATH), "libefusion.so")
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "efusion_replay")
    assert os.path.exists(shim) and os.path.exists(exe)
    syms = subprocess.run(["nm", "-DC", shim], stdout=subprocess.PIPE, text=True).stdout
    for name in ("efusion::ElasticFusion::processFrame(", "efusion::ElasticFusion::predict()", "efusion::ElasticFusion::get_T_wc()",
                 "efusion::ElasticFusion::savePly()", "Resolution::getInstance(int, int)", "Intrinsics::getInstance(float, float, float, float)",
                 "efusion::GlobalModelView::lastCount()"):
        assert name in syms, name
    assert subprocess.run([exe]).returncode == 2
    rgb = np.full((480, 640, 3), 7, np.uint8)
    depth = np.full((480, 640), 1000, np.uint16)
    log = str(tmp_path / "two.klg")
    synth.write_klg(log, [(rgb, depth), (rgb, depth)], compress_depth=True)
    assert os.path.getsize(log) < 4 + 2 * (16 + 640 * 480 * 5)
    if not os.path.exists("/dev/kfd"):
        r = subprocess.run([exe, "-l", log, "-q"], stderr=subprocess.PIPE, text=True)
        assert r.returncode == 1 and "libefusion_hip error" in r.stderr
