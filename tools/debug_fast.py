"""GPU box: the fast order's two tracker scripts (persistent launch / one launch per step) against the oracle, frame by frame."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import efo
from elasticfusion_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["perstep", "persistent"]
seq = synth.Sequence(0xEF0001)
efo.set_threads(16)
o = efo.Fusion()
ost = []
for k in range(n):
    rgb, depth, _ = seq.frame(k)
    o.process_frame(rgb, depth, k * 33333)
    ost.append((np.asarray(o.stats(), np.float32).copy(), o.pose().copy(), o.map_count()))
for mode in modes:
    ef = api.ElasticFusion()
    ef.setPersistentTracker(mode == "persistent")
    bad = 0
    t0 = time.time()
    for k in range(n):
        rgb, depth, _ = seq.frame(k)
        ef.processFrame(rgb, depth, k * 33333)
        st = np.asarray(ef.trackingStats()[0], np.float32)
        T = ef.get_T_wc()
        same = np.array_equal(st.view(np.uint32), ost[k][0].view(np.uint32)) and np.array_equal(T.astype(np.float32), ost[k][1].astype(np.float32))
        bad += not same
        print(f"[{mode}] frame {k}: {'SAME' if same else 'DIFF'} stats {st} vs {ost[k][0]} dT {np.abs(T - ost[k][1]).max():.3e} count {ef.lastCount()} vs {ost[k][2]}", flush=True)
    try:
        ef.synchronize()
        print(f"[{mode}] synchronize ok, {time.time() - t0:.2f} s, {bad} frames differ", flush=True)
    except Exception as e:
        print(f"[{mode}] synchronize: {e}", flush=True)
    ef.close()
