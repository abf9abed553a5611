"""The drop-in claim at compile level (VERDICT r2 missing #6 / next #8).

north_star: "keeping the libefusion.so ElasticFusion::processFrame() C++ API surface so it drops in behind the existing Tools front-end".
Two checks, both on the CPU:

  1. CENSUS — every `eFusion->member` (and what is chained behind the facade getters) that the reference's own MainController.cpp uses is
     read out of /root/reference/MainController.cpp and must either be declared by include/ElasticFusion.h or be on the allow-list of
     members that only exist with OpenGL behind them (textures, feedback buffers, the draw passes: SURVEY §2 marks the display path out
     of scope).  The allow-list is the list INTEGRATION.md prints; the test fails if the reference needs anything beyond it.
  2. COMPILE — tests/front_end/main_controller_calls.cpp spells every non-GL use the census finds the way the reference spells it
     (constructor with its sixteen arguments in order, the run loop, the statistics, the setters, savePly) and is compiled against the
     header with g++ -fsyntax-only, with the POD pose type and with Sophus' type (the oracle's miniature <sophus/se3.hpp>).

Compiling MainController.cpp itself would need Pangolin's GUI, the reference's GPUTexture / Shader classes and a full Eigen (none of them
in this image: the checkout's third-party/ is empty); the uses of class ElasticFusion are what the boundary is about, and those are
compiled."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/MainController.cpp"
HEADER = os.path.join(ROOT, "include", "ElasticFusion.h")
CALLS = os.path.join(ROOT, "tests", "front_end", "main_controller_calls.cpp")

# members of class ElasticFusion (and of the objects its getters return) that are OpenGL objects or draw passes in the reference and have
# no counterpart in a library without a GL context — the lines of MainController.cpp that must change (INTEGRATION.md §"front-end")
GL_ONLY = {
    "computeFeedbackBuffers": "FeedbackBuffer::compute (raw-frame point clouds for display)",
    "getFeedbackBuffers": "std::map<std::string, FeedbackBuffer*>: transform-feedback VBOs drawn by the GUI",
    "normaliseDepth": "display pass writing the DEPTH_NORM texture",
    "getTextures": "std::map<std::string, GPUTexture*>: GL textures shown in the side panels",
    "getGlobalModel.model": "the surfel VBO handed to the GUI's FXAA renderer",
    "getGlobalModel.renderPointCloud": "draw call",
    "getIndexMap.renderDepth": "display pass",
    "getIndexMap.imageTex": "GPUTexture* of the predicted image (the data is available as getIndexMap().image())",
    "getIndexMap.drawTex": "GPUTexture* of the rendered depth",
}


def reference_uses():
    src = open(REF).read()
    uses = set()
    for m in re.finditer(r"eFusion->(\w+)\(\)?((?:\s*\.\s*\w+)*)", src):
        top = m.group(1)
        uses.add(top)
        chain = re.findall(r"\.\s*(\w+)", m.group(2))
        if chain and top in ("getGlobalModel", "getIndexMap", "getModelToModel", "getFerns", "getLocalDeformation"):
            uses.add(top + "." + chain[0])
    return uses


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference checkout is only present in the build container")
def test_every_member_the_reference_front_end_uses_is_declared_or_gl_only():
    header = re.sub(r"//[^\n]*", "", open(HEADER).read())   # declarations only: comments name the reference's members freely
    calls = open(CALLS).read()
    uses = reference_uses()
    assert {"processFrame", "predict", "getTick", "setTick", "get_T_wc", "savePly", "getFerns.frames", "getModelToModel.lastICPError"} <= uses, uses
    missing = []
    for u in sorted(uses):
        if u in GL_ONLY:
            continue
        name = u.split(".")[-1]
        declared = re.search(r"\b%s\b\s*(\(|=|;|,)" % re.escape(name), header) is not None
        exercised = re.search(r"\b%s\b" % re.escape(name), calls) is not None
        if not (declared and exercised):
            missing.append((u, declared, exercised))
    assert not missing, missing
    # the allow-list does not rot: every entry is really used by the reference and really absent from the header
    for u in GL_ONLY:
        assert u in uses, u
        assert re.search(r"\b%s\b\s*\(" % re.escape(u.split(".")[-1]), header) is None, u


@pytest.mark.parametrize("sophus", [False, True], ids=["pod_pose", "sophus_pose"])
def test_the_front_end_calls_compile_against_the_header(sophus):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-unused-variable", "-I" + os.path.join(ROOT, "include")]
    if sophus:
        cmd += ["-DEFUSION_USE_SOPHUS", "-I" + os.path.join(ROOT, "oracle", "host_on_cpu")]
    r = subprocess.run(cmd + [CALLS], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
