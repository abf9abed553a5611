#!/usr/bin/env python
"""Side measurement, NOT the headline metric (bench.py: one replay per GPU): M independent sequences sharing ONE MI355X — M contexts,
each with its own stream and its own host thread, replaying 640x480 frames resident in HBM.  One replay is a chain of dependent
small kernels and keeps well under a tenth of the chip busy, so a server that owns more streams than GPUs
can put several on one device; this prints the aggregate frames/s for M = 1, 2, 4, 8 (one JSON line per M).

    python tools/shared_gpu_bench.py [--steps 150] [--sequences 1,2,4,8]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the synthetic frame generator and its frame cache)

PREROLL = bench.PREROLL


def main():
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(420, exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--sequences", default="1,2,4,8")
    ap.add_argument("--frames-cache", default=None)
    a = ap.parse_args()
    Ms = [int(x) for x in a.sequences.split(",")]
    n_frames = 1 + PREROLL + a.steps
    cache = f"{a.frames_cache}.shared.{n_frames}.npz" if a.frames_cache else None
    if cache and os.path.exists(cache):
        z = np.load(cache)
        frames = [(z["rgb"][k], z["depth"][k]) for k in range(n_frames)]
    else:
        frames = [(r, d) for r, d, _ in bench.generate_frames(0xEF0001, n_frames)]
        if cache:
            np.savez(cache, rgb=np.stack([f[0] for f in frames]), depth=np.stack([f[1] for f in frames]))
    from elasticfusion_amd import api
    dev = [(api.DevBuf.from_array(r), api.DevBuf.from_array(d)) for r, d in frames]

    for M in Ms:
        ctxs = [api.ElasticFusion() for _ in range(M)]          # each creates its own non-blocking stream
        if M > 1:   # the persistent tracker launch takes the whole chip for ~0.4 ms per frame and the launches of a device are chained: several
            for ef in ctxs:   # sequences on one GPU overlap better as launch-per-step scripts (same results)
                ef.setPersistentTracker(False)
        go = threading.Barrier(M + 1)
        done = threading.Barrier(M + 1)

        def worker(ef):
            for k in range(1 + PREROLL):                         # the map reaches its steady state, untimed
                ef.processFrameDevice(dev[k][0].p.value, dev[k][1].p.value, k * 33333)
            ef.synchronize()
            go.wait()
            for k in range(1 + PREROLL, n_frames):
                ef.processFrameDevice(dev[k][0].p.value, dev[k][1].p.value, k * 33333)
            ef.synchronize()
            done.wait()

        th = [threading.Thread(target=worker, args=(c,)) for c in ctxs]
        for t in th:
            t.start()
        go.wait()
        t0 = time.perf_counter()
        done.wait()
        dt = time.perf_counter() - t0
        for t in th:
            t.join()
        poses = [c.get_T_wc() for c in ctxs]
        same = all(np.array_equal(poses[0], p) for p in poses)   # the same frames through independent contexts: the same answer, bit for bit
        counts = [c.lastCount() for c in ctxs]
        for c in ctxs:
            c.close()
        print(json.dumps({"sequences_on_one_gpu": M, "frames_per_s_aggregate": round(M * a.steps / dt, 1), "frames_per_s_per_sequence": round(a.steps / dt, 1),
                          "steps_per_sequence": a.steps, "identical_results": bool(same and len(set(counts)) == 1), "surfels": counts[0]}), flush=True)
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    main()
