"""Developer tool (GPU box): stage timing inside k_solve_update via wall_clock64 stamps (EF_STAGE_CLOCKS build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elasticfusion_amd import api, synth
seq = synth.Sequence(0xEF0001)
ef = api.ElasticFusion()
for k in range(6):
    rgb, depth, _ = seq.frame(k)
    ef.processFrame(rgb, depth, k)
    out = (C.c_ulonglong * 16)()
    api.lib().ef_debug_clocks(ef.h, out)
    v = np.array([int(x) & ((1 << 62) - 1) for x in out], np.int64)
    if k:
        order = [0, 1, 2, 9, 10, 3, 4, 5, 6, 7, 8]
        names = ["wg-start", "rows+chains+store", "ticket8+blocktree", "ticket64", "finaltree", "slots+stats", "A,b", "ldlt", "rodrigues",
                 "mul+compose", "krk"]
        tt = v[order]
        d = np.diff(tt) * 10
        print(f"   first WG start .. last WG start {(v[14]-v[13])*10} ns; first WG start .. end of update {(v[8]-v[13])*10} ns")
        print(f"frame {k}: last-WG total {(tt[-1]-tt[0])*10} ns | " + " ".join(f"{n}={x}" for n, x in zip(names[1:], d)))
