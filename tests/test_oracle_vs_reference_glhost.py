"""The reference's map-side HOST code as its own witness: Core/IndexMap.cpp, GlobalModel.cpp and Shaders/{FillIn,ComputePack,
FeedbackBuffer,Resize}.cpp are compiled from /root/reference where they lie against oracle/host_on_cpu, in which OpenGL and
Pangolin are a tape recorder (gl_record.h), and run.  The transcript of what each call asks the GL to do — which shader files
make the program, every uniform by name with its value, which texture sits on which unit, the attachments and their order, the
viewport, the vertex attribute layout, the draw calls and their order — is compared with what the oracle and
oracle/ref_glsl_bridge.cpp (which drive the compiled shaders by hand) assume for the same pass.  The shaders themselves are pinned
in tests/test_oracle_vs_reference_glsl.py; this pins the parameters they run with."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libefr_glhost.so")
W, H, FX, FY, CX, CY = 640, 480, 528.0, 530.0, 320.0, 240.0      # fx != fy on purpose


def have():
    if not os.path.exists(SO) and os.path.isdir("/root/reference/Core/Shaders"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refglhost"])
    return os.path.exists(SO)


pytestmark = pytest.mark.skipif(not have(), reason="oracle/_ref/libefr_glhost.so can only be built where /root/reference exists")
P = C.c_void_p


class Pass:
    def __init__(self, program):
        self.program, self.uniforms, self.units, self.lines, self.draws, self.attribs = program, {}, {}, [], [], []


def parse(text):
    """-> (preamble lines, [Pass]) : a pass starts at `program Bind:`; state set before it (framebuffer, viewport) is its preamble"""
    passes, pre, unit, cur = [], [], 0, None
    for ln in text.splitlines():
        if ln.startswith("program Bind:"):
            cur = Pass(tuple(ln.split(":", 1)[1].split()))
            cur.pre = pre
            pre = []
            passes.append(cur)
            unit = 0
            continue
        if cur is None or ln.startswith("program Unbind"):
            if ln.startswith("program Unbind"):
                cur = None
            else:
                pre.append(ln)
            continue
        cur.lines.append(ln)
        t = ln.split()
        if t[0] == "uniform":
            kind = t[2]
            vals = [float(x) for x in t[3:] if x.replace("-", "").replace(".", "").replace("e", "").replace("+", "").isdigit()] if kind != "mat4" else [float(x) for x in t[5:]]
            cur.uniforms[t[1]] = vals[0] if len(vals) == 1 else vals
        elif t[0] == "glActiveTexture":
            unit = int(t[1])
        elif t[0] == "glBindTexture" and int(t[2]) != 0:
            cur.units[unit] = int(t[2])
        elif t[0] in ("glDrawArrays", "glDrawTransformFeedback"):
            cur.draws.append(t)
        elif t[0] == "glVertexAttribPointer":
            cur.attribs.append(ln)
    return pre, passes


@pytest.fixture(scope="module")
def host():
    so = C.CDLL(SO)
    so.efh_create.restype = P
    for f in ("take_log", "predict_indices", "combined_predict", "synthesize_depth", "fuse", "clean", "feedback_and_initialise", "fill_in", "resize",
              "compute_packs"):
        getattr(so, "efh_" + f).restype = C.c_char_p
    so.efh_take_log.argtypes = [P]
    so.efh_predict_indices.argtypes = [P, P, C.c_int, C.c_float, C.c_int]
    so.efh_combined_predict.argtypes = [P, P, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
    so.efh_synthesize_depth.argtypes = [P, P, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
    so.efh_fuse.argtypes = [P, P, C.c_int, C.c_float, C.c_float]
    so.efh_clean.argtypes = [P, P, C.c_int, C.c_float, P, C.c_int, C.c_int, C.c_float, C.c_int]
    so.efh_feedback_and_initialise.argtypes = [P, C.c_int, C.c_float]
    so.efh_fill_in.argtypes = [P, C.c_int]
    so.efh_resize.argtypes = [P]
    so.efh_compute_packs.argtypes = [P, C.c_float]
    so.efh_tid.argtypes = [P, C.c_char_p]
    so.efh_constant.argtypes = [C.c_char_p]
    so.efh_model.argtypes = [P, P, P]
    h = P(so.efh_create(W, H, C.c_float(FX), C.c_float(FY), C.c_float(CX), C.c_float(CY)))
    built = so.efh_take_log(h).decode()
    tid = {n: so.efh_tid(h, n.encode()) for n in ("index", "vertConf", "colorTime", "normalRad", "image", "vertex", "normal", "time", "oldImage", "oldVertex",
                                                    "oldNormal", "oldTime", "depth", "rgb", "depthRaw", "depthFiltered", "depthMetric", "depthMetricFiltered",
                                                    "fillImage", "fillVertex", "fillNormal")}
    attach = {}
    for ln in built.splitlines():
        if "AttachColour" in ln:
            t = ln.split()
            attach.setdefault(int(t[1]), []).append(int(t[-1]))
    so.efh_set_query_result(1234)
    return dict(so=so, h=h, tid=tid, attach=attach, built=built, const=lambda n: so.efh_constant(n.encode()))


def pose():
    a = 0.3
    T = np.eye(4)
    T[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
    T[:3, 3] = [0.1, -0.2, 0.3]
    return T


def oracle_mats(T):
    """the oracle's own float matrices (efo_pose.h), as glUniformMatrix4fv receives Eigen's column-major data()"""
    import efo
    a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
    efo.lib().efo_pose_matrices(np.ascontiguousarray(T, np.float64).ctypes.data_as(P), a.ctypes.data_as(P), b.ctypes.data_as(P))
    return a.reshape(4, 4).T.reshape(16), b.reshape(4, 4).T.reshape(16)


def same32(vals, want):
    """%.9g round-trips binary32: the recorded values must equal the oracle's floats exactly"""
    return np.array_equal(np.asarray(vals, np.float64).astype(np.float32), np.asarray(want, np.float32))


def T_cw_colmajor(T):
    return oracle_mats(T)[0]


def bound_fbo(p):
    fb = [ln for ln in p.pre if ln.startswith("GlFramebuffer") and "Bind" in ln]
    return int(fb[-1].split()[1]) if fb else None


SURFEL_LAYOUT = ["glVertexAttribPointer 0 size=4 type=0x1406 norm=0 stride=48 offset=0", "glVertexAttribPointer 1 size=4 type=0x1406 norm=0 stride=48 offset=16",
                 "glVertexAttribPointer 2 size=4 type=0x1406 norm=0 stride=48 offset=32"]


def test_predict_indices(host):
    T = pose()
    _, (p,) = parse(host["so"].efh_predict_indices(host["h"], T.ctypes.data, 7, 20.0, 200).decode())
    t = host["tid"]
    assert p.program == ("index_map.vert", "index_map.frag")
    assert same32(p.uniforms["t_inv"], T_cw_colmajor(T)) and "transpose=0" in [l for l in p.lines if "t_inv" in l][0]
    assert p.uniforms["cam"] == [CX, CY, FX, FY]
    assert (p.uniforms["maxDepth"], p.uniforms["cols"], p.uniforms["rows"], p.uniforms["time"], p.uniforms["timeDelta"]) == (20.0, W, H, 7, 200)
    assert host["attach"][bound_fbo(p)] == [t["index"], t["vertConf"], t["colorTime"], t["normalRad"]]
    assert f"glViewport 0 0 {W} {H}" in p.pre and "glClearColor 0 0 0 0" in p.pre      # IndexMap::FACTOR == 1
    assert p.attribs == SURFEL_LAYOUT and len(p.draws) == 1 and p.draws[0][0] == "glDrawTransformFeedback"


@pytest.mark.parametrize("inactive", [0, 1])
def test_combined_predict(host, inactive):
    T = pose()
    txt = host["so"].efh_combined_predict(host["h"], T.ctypes.data, 20.0, 10.0, 0 if inactive else 9, 5 if inactive else 9, 4, inactive).decode()
    _, (p,) = parse(txt)
    t = host["tid"]
    assert p.program == ("splat.vert", "combo_splat.frag")
    assert same32(p.uniforms["t_inv"], T_cw_colmajor(T)) and p.uniforms["cam"] == [CX, CY, FX, FY]
    u = p.uniforms
    assert (u["maxDepth"], u["confThreshold"], u["cols"], u["rows"], u["timeDelta"]) == (20.0, 10.0, W, H, 4)
    assert (u["time"], u["maxTime"]) == ((0, 5) if inactive else (9, 9))
    want = ["oldImage", "oldVertex", "oldNormal", "oldTime"] if inactive else ["image", "vertex", "normal", "time"]
    assert host["attach"][bound_fbo(p)] == [t[n] for n in want]
    c = host["const"]
    assert f"glEnable {c('GL_PROGRAM_POINT_SIZE'):#x}" in txt and f"glEnable {c('GL_POINT_SPRITE'):#x}" in txt     # sprites sized by the vertex shader
    assert p.attribs == SURFEL_LAYOUT and [d[0] for d in p.draws] == ["glDrawTransformFeedback"]


def test_synthesize_depth(host):
    T = pose()
    _, (p,) = parse(host["so"].efh_synthesize_depth(host["h"], T.ctypes.data, 20.0, 10.0, 9, 5, 4).decode())
    assert p.program == ("splat.vert", "depth_splat.frag") and host["attach"][bound_fbo(p)] == [host["tid"]["depth"]]
    assert (p.uniforms["time"], p.uniforms["maxTime"], p.uniforms["timeDelta"], p.uniforms["confThreshold"]) == (9, 5, 4, 10.0)


def test_fuse(host):
    T = pose()
    _, (data, update) = parse(host["so"].efh_fuse(host["h"], T.ctypes.data, 7, 20.0, 0.8).decode())
    t = host["tid"]
    assert data.program == ("data.vert", "data.geom", "data.frag") and update.program == ("update.vert",)
    u = data.uniforms
    sam = {n: int(u[n]) for n in ("cSampler", "drSampler", "drfSampler", "indexSampler", "vertConfSampler", "colorTimeSampler", "normRadSampler")}
    # colour, RAW metric depth, FILTERED metric depth, then the four index-map images
    assert {n: data.units[k] for n, k in sam.items()} == dict(cSampler=t["rgb"], drSampler=t["depthMetric"], drfSampler=t["depthMetricFiltered"],
                                                               indexSampler=t["index"], vertConfSampler=t["vertConf"], colorTimeSampler=t["colorTime"],
                                                               normRadSampler=t["normalRad"])
    assert same32(u["cam"], [CX, CY, np.float32(1.0 / FX), np.float32(1.0 / FY)])       # inverse focal lengths here
    assert (u["cols"], u["rows"], u["scale"], u["texDim"], u["maxDepth"], u["time"]) == (W, H, 1.0, 3072.0, 20.0, 7.0)
    assert abs(u["weighting"] - np.float32(0.8)) < 1e-9
    assert same32(u["pose"], oracle_mats(T)[1])                           # T_wc.cast<float>().matrix(), not its inverse
    assert "glViewport 0 0 3072 3072" in data.pre                                                           # the update map is TEXTURE_DIMENSION square
    assert data.attribs == ["glVertexAttribPointer 0 size=2 type=0x1406 norm=0 stride=0 offset=0"]         # one vec2 per pixel: the uv buffer
    assert data.draws == [["glDrawArrays", "0", "0", str(W * H)]]
    assert (update.uniforms["texDim"], update.uniforms["time"]) == (3072.0, 7)
    assert [update.units[int(update.uniforms[n])] for n in ("vertSamp", "colorSamp", "normSamp")] == host["attach"][bound_fbo(data)]
    assert update.attribs == SURFEL_LAYOUT and [d[0] for d in update.draws] == ["glDrawTransformFeedback"]


def test_clean(host):
    T = pose()
    g = np.zeros((3, 16), np.float32)
    txt = host["so"].efh_clean(host["h"], T.ctypes.data, 7, 10.0, g.ctypes.data, 3, 200, 20.0, 0).decode()
    pre, (p,) = parse(txt)
    t = host["tid"]
    assert p.program == ("copy_unstable.vert", "copy_unstable.geom")
    u = p.uniforms
    assert (u["time"], u["confThreshold"], u["scale"], u["nodes"], u["nodeCols"], u["timeDelta"], u["maxDepth"], u["isFern"]) == (7, 10.0, 1.0, 3.0, 16384.0, 200, 20.0, 0)
    assert same32(u["t_inv"], T_cw_colmajor(T)) and u["cam"] == [CX, CY, FX, FY] and (u["cols"], u["rows"]) == (W, H)
    want = dict(indexSampler="index", vertConfSampler="vertConf", colorTimeSampler="colorTime", normRadSampler="normalRad", depthSampler="depth")
    assert {n: p.units[int(u[n])] for n in want} == {n: t[v] for n, v in want.items()}
    assert any(l.startswith("glTexSubImage2D") and " 48 1 " in l for l in p.pre)          # the graph: nodes x 16 floats in one row of the node texture
    assert len(p.draws) == 2 and all(d[0] == "glDrawTransformFeedback" for d in p.draws)   # the old map first, then the new unstable surfels
    assert p.attribs == SURFEL_LAYOUT * 2 and "-> 1234" in txt                              # the new count is the primitives-written query


def test_first_frame_seeding(host):
    _, (raw, flt, init) = parse(host["so"].efh_feedback_and_initialise(host["h"], 1, 20.0).decode())
    t = host["tid"]
    for p, depth in ((raw, "depthMetric"), (flt, "depthMetricFiltered")):
        assert p.program == ("vertex_feedback.vert", "vertex_feedback.geom")
        u = p.uniforms
        assert same32(u["cam"], [CX, CY, np.float32(1.0 / FX), np.float32(1.0 / FY)])
        assert (u["cols"], u["rows"], u["time"], u["maxDepth"], u["threshold"]) == (W, H, 1, 20.0, 0.0)
        assert p.units[int(u["gSampler"])] == t[depth] and p.units[int(u["cSampler"])] == t["rgb"]
        assert p.draws == [["glDrawArrays", "0", "0", str(W * H)]]
    # init_unstable.vert: position + colour from the RAW stream, normal + radius from the FILTERED one
    assert init.program == ("init_unstable.vert",) and init.attribs == SURFEL_LAYOUT
    binds = [l for l in init.lines if l.startswith("glBindBuffer 0x8892") and not l.endswith(" 0")]
    attr_after = {}
    cur = None
    for l in init.lines:
        if l.startswith("glBindBuffer 0x8892"):
            cur = l.split()[-1]
        elif l.startswith("glVertexAttribPointer"):
            attr_after[int(l.split()[1])] = cur
    assert attr_after[0] == attr_after[1] != attr_after[2] and len(binds) == 2


def test_fill_in_resize_and_compute_packs(host):
    t = host["tid"]
    _, (fv, fn, fc) = parse(host["so"].efh_fill_in(host["h"], 0).decode())
    for p, frag, e, r in ((fv, "fill_vertex.frag", "vertex", "depthFiltered"), (fn, "fill_normal.frag", "normal", "depthFiltered"), (fc, "fill_rgb.frag", "image", "rgb")):
        assert p.program == ("empty.vert", "quad.geom", frag) and p.uniforms["passthrough"] == 0
        assert p.units[int(p.uniforms["eSampler"])] == t[e] and p.units[int(p.uniforms["rSampler"])] == t[r]
        assert p.draws == [["glDrawArrays", "0", "0", "1"]]
    assert same32(fv.uniforms["cam"], [CX, CY, np.float32(1.0 / FX), np.float32(1.0 / FY)]) and "cam" not in fc.uniforms
    assert [host["attach"][bound_fbo(p)] for p in (fv, fn, fc)] == [[t["fillVertex"]], [t["fillNormal"]], [t["fillImage"]]]
    _, rs = parse(host["so"].efh_resize(host["h"]).decode())
    assert [p.units[0] for p in rs] == [t["image"], t["vertex"], t["oldTime"]] and all(f"glViewport 0 0 {W // 20} {H // 20}" in p.pre for p in rs)
    txt = host["so"].efh_compute_packs(host["h"], 3.0).decode()
    _, (bil, m1, m2) = parse(txt)
    assert bil.program[-1] == "depth_bilateral.frag" and bil.uniforms == dict(cols=W, rows=H, maxD=3.0)
    assert m1.program[-1] == m2.program[-1] == "depth_metric.frag" and m1.uniforms == m2.uniforms == dict(maxD=3.0)
    assert [host["attach"][bound_fbo(p)] for p in (bil, m1, m2)] == [[t["depthFiltered"]], [t["depthMetric"]], [t["depthMetricFiltered"]]]
    srcs = [int(l.split()[-1]) for p in (bil, m1, m2) for l in p.pre if l.startswith("glBindTexture")]
    assert srcs == [t["depthRaw"], t["depthRaw"], t["depthFiltered"]]
