#!/usr/bin/env python
"""Per-phase clocks of the persistent small-level tracker launch (k_track_small), from a -DEF_STAGE_CLOCKS build:

    python -m elasticfusion_amd.build --variant clocks -DEF_STAGE_CLOCKS
    python tools/small_clocks.py elasticfusion_amd/libefusion_hip_clocks.so [more libraries ...] [frames]

Workgroup 0 stamps wall_clock64() (100 MHz) at every phase boundary of every launch; the sums are divided by the launch / iteration
counts here.  The stamps' own global read-modify-writes add ~0.1-0.2 us per stamp."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elasticfusion_amd import api, synth

n = 140
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
           "so3_per_iteration_us": {"rows_chains_publish": us(v[1] / so3_its), "barrier": us(v[2] / so3_its), "gather_tree": us(v[3] / so3_its),
                                    "update": us(v[4] / so3_its)},
           "se3_per_iteration_us": {"head_gather_trees": us(v[5] / se3_its), "head_solve": us(v[6] / se3_its), "search_arrive_A": us(v[7] / se3_its),
                                    "icp_accumulation": us(v[8] / se3_its), "join_trees_publish": us(v[9] / se3_its), "barrier_B": us(v[10] / se3_its),
                                    "rgb_wave_loads_and_wait_A": us(v[12] / se3_its), "rgb_wave_rows_after_A": us(v[13] / se3_its)}}
    print(json.dumps(rec), flush=True)
    ef.close()
