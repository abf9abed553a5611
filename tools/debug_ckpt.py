"""GPU box: the checkpoint / resume scenario of tests/test_gpu_one_frame.py with a per-step witness and explicit synchronisation."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from elasticfusion_amd import api, synth

seq = synth.Sequence(0xEF0002)
frames = [seq.frame(k) for k in range(50)]
for sync_each in (False, True):
    for pa, pb in ((True, True), (False, False), (True, False)):
        a = api.ElasticFusion(); a.setPersistentTracker(pa)
        for k in range(40):
            a.processFrame(frames[k][0], frames[k][1], k * 33333)
        ck = a.checkpoint(frames[39][0], frames[39][1])
        b = api.ElasticFusion(); b.setPersistentTracker(pb); b.restore(ck)
        diffs = []
        for k in range(40, 50):
            for ef in (a, b):
                ef.processFrame(frames[k][0], frames[k][1], k * 33333)
                if sync_each:
                    ef.synchronize()
            qa, qb = a.getPoseQT(), b.getPoseQT()
            sa, sb = np.asarray(a.trackingStats()[0], np.float32), np.asarray(b.trackingStats()[0], np.float32)
            if not (np.array_equal(qa, qb) and np.array_equal(sa.view(np.uint32), sb.view(np.uint32))):
                diffs.append((k, float(np.abs(qa - qb).max()), sa.tolist(), sb.tolist()))
        msg = []
        for name, ef in (("a", a), ("b", b)):
            try:
                ef.synchronize(); msg.append(name + " ok")
            except Exception as e:
                msg.append(name + " " + str(e)[:80])
        print(f"sync_each={sync_each} persistent a={pa} b={pb}: {len(diffs)} frames differ {[d[:2] for d in diffs[:3]]} | {msg}", flush=True)
        if diffs:
            print("   first:", diffs[0], flush=True)
        a.close(); b.close()
