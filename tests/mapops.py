"""Shared harness of the map-side parity tests (the GLSL passes G1-G11 of SURVEY.md §8a): one fixed set of inputs taken
from an oracle run, and one function that pushes them through every pass on any backend with the oracle's Python
signature (tests/efo.py): the oracle, its no-FMA build, the REFERENCE's own shaders compiled for the CPU
(efo.backend("reference_glsl"), oracle/_ref/libefr_glsl.so) or the HIP kernels (HipMapOps(api)).
Every pass gets its inputs from the fixture, never from another pass of the same backend, so a difference cannot cascade.
TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import numpy as np

MAXD = 20.0
TD = 2147483647 // 2
CONF = 1.0


def make_inputs(width=640, height=480, seed=0xEF0002, frames=6):
    """Oracle run (no-FMA build for the map passes == the compiled shaders, see test_oracle_vs_reference_glsl.py) with
    ground-truth poses injected and a low confidence threshold, so that after a few frames the map holds stable, merged,
    cleaned and new unstable surfels.  Returns the arrays every pass needs."""
    import efo
    from elasticfusion_amd import synth
    seq = synth.Sequence(seed, width=width, height=height)
    fr = [seq.frame(k) for k in range(frames)]
    f = efo.Fusion(width=width, height=height, fx=seq.fx, fy=seq.fy, cx=seq.cx, cy=seq.cy, confidence=CONF)
    for k in range(frames - 1):
        f.process_frame(fr[k][0], fr[k][1], k, T_wc=None if k == 0 else fr[k][2])
    cam = efo.make_cam(width, height, seq.fx, seq.fy, seq.cx, seq.cy)
    rgb, depth, T = fr[frames - 1]
    tick = f.tick()
    surf = f.map()
    depth0 = fr[0][1].copy()
    h, w = depth0.shape
    depth0[h // 5:h // 5 + h // 12, w // 3:w // 3 + w // 10] = 0            # holes
    depth0[5 * h // 8:5 * h // 8 + 3, w // 12:5 * w // 8] = 305            # near the 300 mm gate
    depth0[5 * h // 8 + 3:5 * h // 8 + 6, w // 12:5 * w // 8] = 296
    with efo.backend("nofma"):
        dm0 = efo.metricise_depth(depth0, 3.0)
        dmf0 = efo.metricise_depth(efo.filter_depth(depth0, 3.0), 3.0)
        dm = efo.metricise_depth(depth, 3.0)
        dmf = efo.metricise_depth(efo.filter_depth(depth, 3.0), 3.0)
        idx, vc, ct, nr = efo.predict_indices(cam, T, tick, surf, MAXD, TD)
        img, vt, nm, tm = efo.combined_predict(cam, T, surf, MAXD, CONF, tick, tick, TD)
        s2, nu = efo.fuse(cam, T, tick, rgb, dm, dmf, idx, vc, ct, nr, MAXD, 0.8, surf)
        idx2, vc2, ct2, nr2 = efo.predict_indices(cam, T, tick, s2, MAXD, TD)
        synth = efo.synthesize_depth(cam, T, surf, MAXD, CONF, tick, tick, TD)
    inp = dict(cam=np.array([width, height, seq.fx, seq.fy, seq.cx, seq.cy], np.float64), T=np.asarray(T, np.float64), tick=np.int32(tick),
               raw=fr[1][1], rgb0=fr[0][0], dm0=dm0, dmf0=dmf0, surf=surf, rgb=rgb, rgb_prev=fr[frames - 2][0], dm=dm, dmf=dmf,
               depth_filtered=f.buffer("depthFiltered"), idx=idx, vc=vc, ct=ct, nr=nr, img=img, vt=vt, nm=nm, s2=s2, nu=nu,
               idx2=idx2, vc2=vc2, ct2=ct2, nr2=nr2, synth=synth)
    return {k: np.ascontiguousarray(v) for k, v in inp.items()}


def make_graph(inp, n_nodes=48, seed=7):
    """A deformation graph in the reference's node-texture layout (GlobalModel.cpp:540-546): per node {position 3,
    rotation 9 column-major, translation 3, time}, sorted by time.  Nodes sit on surfels of the map (as
    Deformation::sampleGraphModel picks them), rotations a few degrees off identity, translations a few centimetres."""
    rng = np.random.RandomState(seed)
    surf = inp["surf"]
    pick = np.sort(rng.choice(len(surf), n_nodes, replace=False))
    g = np.zeros((n_nodes, 16), np.float32)
    g[:, 0:3] = surf[pick, 0:3]
    for i in range(n_nodes):
        w = rng.uniform(-0.05, 0.05, 3)
        th = np.linalg.norm(w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)
        g[i, 3:12] = R.T.reshape(9)          # column-major
        g[i, 12:15] = rng.uniform(-0.03, 0.03, 3)
    tick = int(_scalar(inp["tick"]))
    g[:, 15] = np.sort(rng.randint(0, max(tick, 2), n_nodes)).astype(np.float32)
    return g


def run_deform(be, inp, graph, cam=None, isFern=0):
    """GlobalModel::clean with the deformation graph applied, as after a loop closure (ElasticFusion.cpp:561-585):
    depth = synthesizeDepth of the surfels outside the time window."""
    c = inp["cam"]
    if cam is None:
        cam = be.make_cam(int(c[0]), int(c[1]), float(c[2]), float(c[3]), float(c[4]), float(c[5]))
    T, tick = inp["T"].reshape(4, 4), int(_scalar(inp["tick"]))
    return be.clean_deform(cam, T, tick, inp["idx2"], inp["vc2"], inp["ct2"], inp["nr2"], CONF, TD, MAXD, inp["s2"], inp["nu"], graph,
                           inp["synth"], isFern)


def _scalar(x):
    return np.asarray(x).reshape(-1)[0]


def run_passes(be, inp, cam=None):
    """All map passes on backend `be` (efo-style module or HipMapOps); `cam` = that backend's camera struct."""
    c = inp["cam"]
    if cam is None:
        cam = be.make_cam(int(c[0]), int(c[1]), float(c[2]), float(c[3]), float(c[4]), float(c[5]))
    T, tick = inp["T"].reshape(4, 4), int(_scalar(inp["tick"]))
    out = {}
    flt = be.filter_depth(inp["raw"], 3.0)
    out["filter_depth"] = flt
    out["metric_raw"] = be.metricise_depth(inp["raw"], 3.0)
    out["metric_filtered"] = be.metricise_depth(flt, 3.0)
    out["seed_map"] = be.seed_map(cam, inp["rgb0"], inp["dm0"], inp["dmf0"], 1, MAXD)
    out["index"], out["index_vertConf"], out["index_colorTime"], out["index_normRad"] = be.predict_indices(cam, T, tick, inp["surf"], MAXD, TD)
    out["predict_image"], out["predict_vertex"], out["predict_normal"], out["predict_time"] = be.combined_predict(
        cam, T, inp["surf"], MAXD, CONF, tick, tick, TD)
    # synthesizeDepth as ElasticFusion.cpp:561-570 calls it: only surfels outside the time window (maxTime = time - timeDelta)
    out["synth_depth"] = be.synthesize_depth(cam, T, inp["surf"], MAXD, CONF, tick, tick, TD)
    out["synth_depth_old"] = be.synthesize_depth(cam, T, inp["surf"], MAXD, CONF, tick, tick - 2, TD)
    out["fill_image"], out["fill_vertex"], out["fill_normal"] = be.fill_in(cam, inp["img"], inp["vt"], inp["nm"], inp["depth_filtered"],
                                                                           inp["rgb_prev"])
    out["passthrough_image"], out["passthrough_vertex"], out["passthrough_normal"] = be.fill_in(
        cam, inp["img"], inp["vt"], inp["nm"], inp["depth_filtered"], inp["rgb_prev"], 1, 1)
    out["fuse_map"], out["fuse_new_unstable"] = be.fuse(cam, T, tick, inp["rgb"], inp["dm"], inp["dmf"], inp["idx"], inp["vc"], inp["ct"],
                                                        inp["nr"], MAXD, 0.8, inp["surf"])
    out["clean_map"] = be.clean(cam, T, tick, inp["idx2"], inp["vc2"], inp["ct2"], inp["nr2"], CONF, TD, MAXD, inp["s2"], inp["nu"])
    return out


INDEX_OUTPUTS = ("index", "index_vertConf", "index_colorTime", "index_normRad")
# fraction of index-map PIXELS that may differ between the shader run and the specification: NONE since round 5 — the specification now
# restates index_map.vert's round trip through NDC (window position floor(((ndc + 1) / 2) * size) on the float NDC value, which moves a
# point sitting within ~1e-5 px of a pixel edge into the neighbouring pixel, N1); rounds 1-4 took floor(u) and tolerated 2e-4 of the pixels
INDEX_PIXEL_TOLERANCE = 0.0


def index_pixels_differing(a, b):
    """pixels at which ANY of the four index-map outputs differ"""
    bad = np.zeros(a["index"].shape, bool)
    for k in INDEX_OUTPUTS:
        x, y = a[k], b[k]
        d = x.view(np.uint32) != y.view(np.uint32)
        bad |= d if d.ndim == 2 else d.any(axis=-1)
    return int(bad.sum()), bad.size


class HipMapOps:
    """api.ops map passes with the oracle's call signatures."""

    def __init__(self, api):
        self._api, self._o = api, api.ops

    def make_cam(self, w, h, fx, fy, cx, cy):
        return self._api.ef_cam(w, h, fx, fy, cx, cy)

    def __getattr__(self, name):
        return getattr(self._o, name)
