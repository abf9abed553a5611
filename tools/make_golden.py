#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ from the REFERENCE's own sources compiled for the CPU:

tracking_ops_reference.npz: inputs + the outputs of the reference's 16 tracking operators (Core/Cuda/{reduce,cudafuncs}.cu,
`make -C oracle ref`, see oracle/cuda_on_cpu/) on those inputs;
map_passes_reference.npz: inputs + the outputs of the reference's 18 hot-path shaders (Core/Shaders/*, `make -C oracle
refglsl`, see oracle/glsl_on_cpu/ and oracle/ref_glsl_bridge.cpp) run pass by pass at 96x72.

Must be run where /root/reference exists (this container); the fixture then pins the oracle and the HIP kernels anywhere
(tests/test_oracle_golden.py, tests/test_gpu_vs_reference.py).  Inputs: pyramid level 2 (160x120) of the tracking state
after 3 frames of synthetic sequence 0xEF0001.

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import efo  # noqa: E402
import trackops  # noqa: E402
from elasticfusion_amd import synth  # noqa: E402


def main():
    assert efo.have_reference(), "needs oracle/_ref/libefr_cuda.so (make -C oracle ref, /root/reference present)"
    seq = synth.Sequence(0xEF0001)
    f = efo.Fusion()
    for k in range(3):
        rgb, depth, _ = seq.frame(k)
        f.process_frame(rgb, depth, k)
    inp = trackops.make_inputs(f, seq.frame(2)[0], level=2)
    with efo.backend("reference"):
        out = trackops.run_ops(efo, inp)
    path = os.path.join(ROOT, "tests", "golden", "tracking_ops_reference.npz")
    np.savez_compressed(path, **{"in_" + k: v for k, v in inp.items()}, **{"out_" + k: v for k, v in out.items()})
    print(path, os.path.getsize(path), "bytes;", len(inp), "inputs,", len(out), "outputs")

    # map side: the reference's own shaders (make -C oracle refglsl), 96x72 so that the fixture stays small
    import mapops
    so = efo.reference_glsl_lib()
    so.efg_use_specified_exp(1)
    so.efg_set_depth_compare(1)
    minp = mapops.make_inputs(96, 72)
    with efo.backend("reference_glsl"):
        mout = mapops.run_passes(efo, minp)
    path = os.path.join(ROOT, "tests", "golden", "map_passes_reference.npz")
    np.savez_compressed(path, **{"in_" + k: v for k, v in minp.items()}, **{"out_" + k: v for k, v in mout.items()})
    print(path, os.path.getsize(path), "bytes;", len(minp), "inputs,", len(mout), "outputs;", len(minp["surf"]), "surfels")


if __name__ == "__main__":
    main()
