"""Developer tool: where the time of the level-0 normal-equation kernel goes.  Needs a library built with -DEF_ACCUM_CLOCKS
(python -m elasticfusion_amd.build --variant clocks -DEF_ACCUM_CLOCKS) selected with EF_HIP_LIB; prints, over the 256 workgroups of
the LAST level-0 launch, the wall_clock64() (100 MHz) stamps of wavefront 0 relative to the earliest entry."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elasticfusion_amd import api, synth  # noqa: E402

seq = synth.Sequence(0xEF0001)
ef = api.ElasticFusion()
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    rgb, depth, _ = seq.frame(k)
    ef.processFrame(rgb, depth, k)
ef.synchronize()
buf = (C.c_ulonglong * (8 * 256))()
assert api.lib().ef_debug_accum_stamps(buf) == 0
t = np.array(buf, dtype=np.float64).reshape(8, 256)[:6] * 0.01   # us
t0 = t[0].min()
names = ["entry", "stage-1 consumed, gathers issued", "gathers consumed, rows done", "outer products issued", "workgroup joined", "exit"]
out = {}
for i, n in enumerate(names):
    r = t[i] - t0
    out[n] = dict(min=round(float(r.min()), 2), median=round(float(np.median(r)), 2), max=round(float(r.max()), 2))
    print(f"{n:36s} min {r.min():6.2f}  median {np.median(r):6.2f}  max {r.max():6.2f} us")
d = np.diff(t, axis=0)
for i in range(5):
    print(f"  {names[i]} -> {names[i + 1]}: median {np.median(d[i]):.2f} us (min {d[i].min():.2f}, max {d[i].max():.2f})")
print(json.dumps(out))
