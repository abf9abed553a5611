import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from elasticfusion_amd import api, synth, build
seq = synth.Sequence(0xEF0001)
frames = [seq.frame(k) for k in range(6)]
libs = sys.argv[1:] or [None]
for lib in libs:
    api.use_library(None if lib in (None, "d") else os.path.join(os.path.dirname(build.LIB), f"libefusion_hip_{lib}.so"))
    ef = api.ElasticFusion()
    out = []
    for k in range(6):
        if k == 3:
            ef.synchronize(); ef.debugOccupy(96, 30000); time.sleep(0.002)
        t0 = time.perf_counter()
        ef.processFrame(frames[k][0], frames[k][1], k * 33333)
        ef.synchronize()
        out.append((k, round(1e3 * (time.perf_counter() - t0), 2), ef.trackerFallbacks()))
    print(lib, out, flush=True)
    ef.close()
