"""Developer tool (GPU box): stage timing inside k_se3_accum (level 0) via wall_clock64 stamps.
Build: EF_HIPCC_FLAGS=-DEF_ACCUM_CLOCKS python -m elasticfusion_amd.build --force  (or point EF_HIP_LIB at such a build)."""
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
        names = ["entry->sigma", "phaseA(stage2+store+sync)", "phaseB chains", "tree+store"]
        for wg, base in (("WG0", 0), ("WG511", 8)):
            d = np.diff(v[base:base + 5]) * 10
            print(f"frame {k} {wg}: total {(v[base+4]-v[base])*10} ns | " + " ".join(f"{n}={x}" for n, x in zip(names, d)))
        print(f"   WG0 entry .. WG511 entry {(v[8]-v[0])*10} ns ; WG0 entry .. WG511 end {(v[12]-v[0])*10} ns")

# every workgroup of the last level-0 launch: entry / exit, grouped by XCD (workgroup id % 8)
if hasattr(api.lib(), "ef_debug_accum_stamps"):
    buf = (C.c_ulonglong * 1024)()
    api.lib().ef_synchronize(ef.h)
    api.lib().ef_debug_accum_stamps(buf)
    v = np.array([int(x) for x in buf], np.int64)
    ent, ext = v[:512], v[512:]
    t0 = ent.min()
    print("entry (us after the first) by XCD: " + " ".join(f"x{x}:{(ent[x::8].min()-t0)/100:.2f}-{(ent[x::8].max()-t0)/100:.2f}" for x in range(8)))
    print("exit                       by XCD: " + " ".join(f"x{x}:{(ext[x::8].min()-t0)/100:.2f}-{(ext[x::8].max()-t0)/100:.2f}" for x in range(8)))
    late = np.sort((ent - t0) / 100.0)
    print("entry quantiles us:", [round(float(late[int(q * 511)]), 2) for q in (0, .25, .5, .75, .9, 1)], " workgroups entering after 2 us:", int((late > 2).sum()))
    print("per-workgroup duration us: min %.2f median %.2f max %.2f ; whole launch %.2f us" % ((ext - ent).min() / 100, np.median(ext - ent) / 100, (ext - ent).max() / 100, (ext.max() - t0) / 100))
