"""Per-frame HIP-vs-oracle divergence hunt (GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import efo
from elasticfusion_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seq = synth.Sequence(0xEF0001)
ef = api.ElasticFusion(); o = efo.Fusion()
ocam = efo.make_cam(640, 480, 528, 528, 320, 240)

def cmpf(a, b):
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb): return f"NaN-mask differs ({(na!=nb).sum()})"
    d = np.abs(a[~na].astype(np.float64) - b[~nb].astype(np.float64))
    return f"max|d|={d.max():.3e} n_diff={(d>0).sum()}" if d.size else "empty"

for k in range(n):
    rgb, depth, Tgt = seq.frame(k)
    dense_o = efo.dense_enough(ocam, o.buffer("image")) if k > 0 else None
    ef.processFrame(rgb, depth, k); o.process_frame(rgb, depth, k)
    T, Tr = ef.get_T_wc(), o.pose()
    dt = np.linalg.norm(T[:3,3]-Tr[:3,3]); dR = T[:3,:3].T@Tr[:3,:3]; ang = np.arccos(np.clip((np.trace(dR)-1)/2,-1,1))
    st = ef.trackingStats()[0]; so = o.stats()
    print(f"frame {k}: dpose {dt:.3e} m {ang:.3e} rad | count {ef.lastCount()} vs {o.map_count()} | dense_o {dense_o} | icp {st[0]:.4e}/{st[1]:.0f} vs {so[0]:.4e}/{so[1]:.0f} | rgb {st[3]:.0f} vs {so[3]:.0f} | so3 {st[4]:.5f}/{st[5]:.0f} vs {so[4]:.5f}/{so[5]:.0f}")
    if k > 0:
        odo = o.odometry()
        for name in ("vmap_g_prev", "nmap_g_prev", "vmap_curr", "nmap_curr", "lastDepth"):
            for l in (0, 2):
                a, b = ef.trackerBuffer(name, l), odo.buffer(name, l)
                if name.startswith(("vmap", "nmap")):
                    h = a.shape[0] // 3
                    bad = np.isnan(b[:h])
                    for arr in (a, b):
                        arr[h:2*h][bad] = 0; arr[2*h:][bad] = 0
                    badg = np.isnan(a[:h])
                    for arr in (a, b):
                        arr[h:2*h][badg] = 0; arr[2*h:][badg] = 0
                print(f"    {name}[{l}]: {cmpf(a, b)}")
        for name in ("lastImage", "nextImage", "dIdx"):
            a, b = ef.trackerBuffer(name, 0), odo.buffer(name, 0)
            print(f"    {name}[0]: n_diff={(a != b).sum()}")
        # note: after tracking with so3 the oracle swapped lastNextImage/nextImage; the HIP side swaps pointers too
    for nm, onm in (("image", "image"), ("vertex", "vertex"), ("fill_vertex", "fill_vertex"), ("fill_normal", "fill_normal")):
        a, b = ef.image(nm), o.buffer(onm)
        print(f"    {nm}: " + (f"n_diff={(a != b).sum()}" if a.dtype == np.uint8 else cmpf(a, b)))
    m, mr = ef.downloadMap(), o.map()
    if m.shape == mr.shape:
        print(f"    map: {cmpf(m, mr)} rows_diff={(m != mr).any(axis=1).sum()}")
