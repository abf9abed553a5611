#!/usr/bin/env python
"""Same-box A/B of library builds / tracker scripts in ONE process (GPU boxes of the pool differ by 10-20 %; a `gpurun` visit is charged by the
minute, and bench.py pays Python + torch start-up and frame generation per run):

    python tools/ab_bench.py [--steps 200] [--reps 2] spec ...

spec = <library>[@<mode>][+ov<N>]: library "d" = libefusion_hip.so (the default: reference rounding), a name = libefusion_hip_<name>.so; mode = the
argument of ef_set_persistent_tracker (1 = one persistent launch, 0 = one launch per step, 2 = round 3's small-level launch); +ov<N> = the next
frame's input stage on a second stream beside this frame's fusion (ef_set_input_overlap(ctx, 1)), restricted to every N-th CU (0 = unmasked).  Every spec
replays the same 640x480 frames (resident in HBM) after bench.py's 100-frame pre-roll; prints frames/s per spec and repetition, then JSON."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("specs", nargs="+")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--preroll", type=int, default=100)
    ap.add_argument("--close-loops", action="store_true")
    ap.add_argument("--big", action="store_true", help="BASELINE configs[2]: 1280x960 on a map pre-seeded with ~1 M surfels (bench.py's side leg: preroll 16, warm-up 24)")
    a = ap.parse_args()
    import bench
    from elasticfusion_amd import api, build
    w, h = (1280, 960) if a.big else (640, 480)
    if a.big:
        a.preroll = 16
    n = 1 + a.preroll + 20 + a.steps
    frames = bench.generate_frames(0xEF0001, n, w, h)
    out = {}
    dev = None
    for rep in range(a.reps):
        for spec in a.specs:
            host = spec.endswith("+host")   # frames handed over as HOST pointers (ef_process_frame: pinned ring + upload on the copy stream)
            if host:
                spec = spec[:-5]
            nores = spec.endswith("+nores")   # round 5's streaming persistent launch (ef_set_resident_levels(ctx, 0))
            spec_ = spec[:-6] if nores else spec
            base, _, ov = spec_.partition("+ov")
            name, _, mode = base.partition("@")
            api.use_library(None if name in ("d", "-") else os.path.join(os.path.dirname(build.LIB), f"libefusion_hip_{name}.so"))
            dev = [(api.DevBuf.from_array(r), api.DevBuf.from_array(d)) for r, d, _ in frames]   # (per library: the allocator is the library's)
            ef = bench.make_engine(api, w, h, 0, 0, close_loops=a.close_loops)
            k0 = 0
            if a.big:
                bench.preseed(ef, 0xEF0001, w, h, 1 << 20, frames[0])
                k0 = 1
            if mode != "":
                ef.setPersistentTracker(int(mode))
            if nores:
                ef.setResidentLevels(False)
            if ov != "":
                if int(ov) > 1:
                    ef.setInputCuMask(int(ov))
                ef.setInputOverlap(1)
            first = 1 + a.preroll + 20
            def step(k):
                if host:
                    ef.processFrame(frames[k][0], frames[k][1], k * 33333)
                else:
                    ef.processFrameDevice(dev[k][0].p.value, dev[k][1].p.value, k * 33333)
            for k in range(k0, first):
                step(k)
            ef.synchronize()
            t0 = time.perf_counter()
            for k in range(first, first + a.steps):
                step(k)
            ef.synchronize()
            dt = time.perf_counter() - t0
            fps = a.steps / dt
            spec = spec + ("+host" if host else "")
            out.setdefault(spec, []).append(round(fps, 1))
            print(f"[{spec}] rep {rep}: {fps:.1f} frames/s ({1e3 * dt / a.steps:.4f} ms/frame), surfels {ef.lastCount()}, fallbacks {ef.trackerFallbacks()}", flush=True)
            ef.close()
            del dev
    print(json.dumps(out))


if __name__ == "__main__":
    main()
