#!/usr/bin/env python
"""Per-phase clocks of the persistent tracker launch (k_track_fast), from a -DEF_STAGE_CLOCKS build:

    python -m elasticfusion_amd.build --variant clocks -DEF_STAGE_CLOCKS
    python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so [frames]

Workgroup 0 stamps wall_clock64() (100 MHz) at every phase boundary of every launch (thread 0 = lane 0 of the first ICP wavefront); the
sums are divided by the launch / iteration counts here."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elasticfusion_amd import api, synth

n = 140
NORES = "--nores" in sys.argv   # round 5's streaming persistent launch (ef_set_resident_levels(ctx, 0))
sys.argv = [a for a in sys.argv if a != "--nores"]
libs = [a for a in sys.argv[1:] if not a.isdigit()] or [None]
for a in sys.argv[1:]:
    if a.isdigit():
        n = int(a)
seq = synth.Sequence(0xEF0001)
frames = [seq.frame(k)[:2] for k in range(n)]
us = lambda x: round(x * 0.01, 3)
for path in libs:
    if path:
        api.use_library(os.path.abspath(path))
    ef = api.ElasticFusion()
    if NORES:
        ef.setResidentLevels(False)
    L = api.lib()
    out = (C.c_ulonglong * 32)()
    for k, (rgb, depth) in enumerate(frames):
        ef.processFrame(rgb, depth, k * 33333)
        if k == n - 41:
            ef.synchronize()
            L.ef_debug_small_clocks(ef.h, out)   # reset: the last 40 frames (mature map) are what is reported
    ef.synchronize()
    L.ef_debug_small_clocks(ef.h, out)
    v = np.array(list(out), np.float64)
    launches, so3_its, se3_its = max(v[22], 1), max(v[20], 1), max(v[21], 1)
    rec = {"library": os.path.basename(api.LIB_PATH), "launches": int(launches), "so3_iterations_per_launch": round(so3_its / launches, 2),
           "se3_iterations_per_launch": round(se3_its / launches, 2), "whole_launch_us": us(v[11] / launches), "begin_us": us(v[0] / launches),
           "so3_per_iteration_us": {"rows_trees_publish": us(v[1] / so3_its), "exchange": us(v[2] / so3_its), "update": us(v[4] / so3_its)},
           "se3_per_iteration_us": {"head_solve": us(v[6] / se3_its), "search_publish_A": us(v[7] / se3_its), "icp_rows_wave0_resident": us(v[8] / se3_its),
                                    "rest_of_tasks_trees_publish_B": us(v[9] / se3_its), "exchange_B": us(v[10] / se3_its)},
           "solve_phases_us": {"A_b": us(v[3] / max(se3_its - launches, 1)), "ldlt": us(v[5] / max(se3_its - launches, 1)), "rodrigues": us(v[18] / max(se3_its - launches, 1)),
                               "compose_Rcurr": us(v[19] / max(se3_its - launches, 1)), "krk": us(v[23] / max(se3_its - launches, 1))},
           "resident_levels": (not NORES),
           "resident_photometric_wavefront_us": {"search": us(v[24] / se3_its), "publish_A_sweep_sigma": us(v[25] / se3_its), "rows": us(v[26] / se3_its)},
           "iteration_us_by_level": {f"L{l}": us(v[12 + l] / max(v[15 + l], 1)) for l in range(3)},
           "iterations_by_level": {f"L{l}": round(v[15 + l] / launches, 2) for l in range(3)}}
    print(json.dumps(rec), flush=True)
    ef.close()
