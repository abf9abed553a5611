"""Ad-hoc GPU timing of the frame pipeline (not the official bench): per-stage hipEvent times + fps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elasticfusion_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seq = synth.Sequence(0xEF0001)
frames = [seq.frame(k) for k in range(n)]
ef = api.ElasticFusion()
# device-resident frames
dev = [(api.DevBuf.from_array(r), api.DevBuf.from_array(d)) for r, d, _ in frames]
ef.processFrameDevice(dev[0][0].p.value, dev[0][1].p.value, 0)
ef.synchronize()
t0 = time.time()
for k in range(1, n):
    ef.processFrameDevice(dev[k][0].p.value, dev[k][1].p.value, k)
t_enq = time.time() - t0
ef.synchronize()
t1 = time.time() - t0
print(f"frames {n-1}: enqueue {t_enq*1e3:.1f} ms, total {t1*1e3:.1f} ms -> {(n-1)/t1:.1f} fps, {t1/(n-1)*1e3:.3f} ms/frame")
print("count", ef.lastCount(), "pose err vs GT", np.linalg.norm(ef.get_T_wc()[:3, 3] - frames[n-1][2][:3, 3]))
print("stats", ef.trackingStats()[0])
ef.enableTiming(True)
ef2 = api.ElasticFusion()
ef2.enableTiming(True)
for k in range(min(n, 20)):
    ef2.processFrameDevice(dev[k][0].p.value, dev[k][1].p.value, k)
ef2.synchronize()
for name, ms in ef2.timings().items():
    print(f"  {name:22s} {ms*1e3:9.1f} us")
