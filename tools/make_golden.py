#!/usr/bin/env python
"""Generates tests/golden/tracking_ops_reference.npz: inputs + the outputs of the REFERENCE's own tracking operators
(Core/Cuda/{reduce,cudafuncs}.cu compiled for the CPU by `make -C oracle ref`, see oracle/cuda_on_cpu/) on those inputs.

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


if __name__ == "__main__":
    main()
