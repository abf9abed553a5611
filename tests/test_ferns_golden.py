"""The product's fern database (ef_ferns_*, host side of libefusion_hip.so) and the oracle's restatement (efo_ferns_*) replaying the session the REFERENCE's own Core/Ferns.cpp
answered in tests/golden/ferns_reference.npz (tools/make_ferns_golden.py): same fern table from the seed, same frames kept, same
codes, same matches, same recovered poses and constraints.  Needs neither /root/reference nor a GPU."""
import os

import numpy as np
import pytest

import efo
from elasticfusion_amd import api, build
from fernscene import CX, CY, FX, FY, H, W, geometry

BACKENDS = {"product": api.Ferns, "oracle": efo.Ferns}
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ferns_reference.npz")


@pytest.mark.parametrize("backend", list(BACKENDS))
def test_session_answers_match_the_compiled_reference(backend):
    build.build()
    Ferns = BACKENDS[backend]
    g = np.load(GOLDEN)
    f = Ferns(500, 3000, 115.0, W, H, FX, FY, CX, CY, seed=int(g["seed"]))
    assert np.array_equal(f.conservatory, g["table"])            # std::mt19937 + uniform_int_distribution in generateFerns' order
    kept = []
    for rgb, z, T, t in zip(g["add_rgb"], g["add_z"], g["add_T"], g["add_time"]):
        verts, norms = geometry(z)
        kept.append(int(f.addFrame(rgb, verts, norms, T, int(t), float(g["threshold"]))))
    assert kept == list(g["add_kept"]) and len(f) == len(g["codes"])
    for i in range(len(f)):
        s = f.frame(i)
        assert np.array_equal(s["codes"], g["codes"][i]) and s["goodCodes"] == g["good"][i] and s["srcTime"] == g["src"][i]
    for rgb, z, (t, lost, err, cnt), delta, closest, Tr, cons, n in zip(g["q_rgb"], g["q_z"], g["q_par"], g["q_delta"], g["q_closest"], g["q_T"], g["q_cons"],
                                                                        g["q_n"]):
        verts, norms = geometry(z)
        T_est, c = f.findFrame(rgb, verts, norms, g["T_cur"], int(t), bool(lost), lambda fv, fn, Tf, cv, cn, Tin: (Tin @ delta, np.float32(err), np.float32(cnt)))
        assert f.lastClosest == closest
        assert np.abs(T_est - Tr).max() < 1e-12
        assert len(c) == n and (n == 0 or np.abs(c - cons[:n]).max() < 1e-12)
    assert (g["q_closest"] >= 0).sum() >= 4 and (g["q_closest"] < 0).sum() >= 4      # the session exercises both outcomes
    f.close()
