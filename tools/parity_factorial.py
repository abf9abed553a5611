"""The parity factorial: which of the two differences between the shipped build and the reference rounding —
fused multiply-adds, summation order — moves the pose of ONE tracked frame from IDENTICAL state, and by how much.

    python tools/parity_factorial.py [out.json]          (GPU box; ~4 minutes)

The donor is the reference-rounding build (libefusion_hip_nofma.so: no FMAs, the reference's summation order; bit for bit the compiled
reference, tests/test_gpu_vs_reference.py), free-running over the 11 runs of tests/test_gpu_one_frame.py; at each of its 113 checkpoints
every build of the 2 x 2 design processes the SAME next frame from the SAME restored state:

    column                  FMAs   order       library
    reference_rounding      no     reference   libefusion_hip_nofma.so        (the donor: difference 0 by construction, asserted)
    fma_reference_order     yes    reference   libefusion_hip_reforder.so     (-DEF_REF_ORDER: round 3's product)
    nofma_fast_order        no     fast        libefusion_hip_nofma_fast.so   (-DEF_NO_FMA -DEF_FORCE_FAST_ORDER)
    fma_fast_order          yes    fast        libefusion_hip_fast.so         (the opt-in fast build; rounds 4's shipped build)

Per checkpoint and column: pose difference to the donor's frame (m, rad) and the build's one-frame MOTION error against the generating
trajectory (the frame-to-frame motion the synthetic sequence was rendered with), so that a difference between two builds can be read next
to the distance both keep from the truth.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_one_frame as H   # noqa: E402  (the harness: runs, frame rendering, restore, pose metrics)


def motion_error(qt_before, qt_after, T_gen_before, T_gen_after):
    """one-frame motion of a build (pose after the frame relative to the restored pose) against the generating motion: (m, rad)"""
    A = np.linalg.inv(H.qt_matrix(qt_before)) @ H.qt_matrix(qt_after)
    G = np.linalg.inv(T_gen_before) @ T_gen_after
    D = np.linalg.inv(G) @ A
    ang = float(np.arccos(np.clip((np.trace(D[:3, :3]) - 1) / 2, -1, 1)))
    return float(np.linalg.norm(D[:3, 3])), ang


def main(out_path):
    import multiprocessing as mp
    from elasticfusion_amd import api, build
    here = os.path.dirname(build.LIB)
    columns = [("fma_reference_order", os.path.join(here, "libefusion_hip_reforder.so")),
               ("nofma_fast_order", os.path.join(here, "libefusion_hip_nofma_fast.so")),
               ("fma_fast_order", build.FAST_LIB if hasattr(build, "FAST_LIB") else build.LIB)]
    for _, p in columns:
        assert os.path.exists(p), p
    recs = []
    with mp.get_context("spawn").Pool(max(1, min(24, (os.cpu_count() or 2) - 1))) as pool:
        for run in H.RUNS:
            seed, noise, scene, w, h, checks = run
            n = max(checks) + 1
            frames = pool.map(H._frame_job, [(seed, noise, scene, w, h, k) for k in range(n)], chunksize=4)
            size = (w, h)
            api.use_library(build.NOFMA_LIB)
            cks, donor = {}, {}
            try:
                ef = H.engine(api, w, h)
                for k, (rgb, depth, _) in enumerate(frames):
                    if k in checks:
                        cks[k] = ef.checkpoint(frames[k - 1][0], frames[k - 1][1])
                    ef.processFrame(rgb, depth, k * 33333)
                    if k in checks:
                        donor[k] = ef.getPoseQT()
                ef.close()
                for k in (checks[0], checks[-1]):   # the restore is complete: the donor's own build reproduces the donor's frame bit for bit
                    assert np.array_equal(H.one_frame(api, cks[k], frames[k], k, size=size)["qt"], donor[k]), (run, k)
            finally:
                api.use_library(None)
            per = {k: dict(seed=hex(seed), noise=bool(noise), scene=scene, size=[w, h], frame=k, builds={}) for k in checks}
            for k in checks:
                em, ea = motion_error(cks[k]["qt"], donor[k], frames[k - 1][2], frames[k][2])
                per[k]["builds"]["reference_rounding"] = dict(pose_difference_m=0.0, pose_difference_rad=0.0, motion_error_m=em, motion_error_rad=ea)
            for name, path in columns:
                api.use_library(path)
                try:
                    for k in checks:
                        got = H.one_frame(api, cks[k], frames[k], k, size=size)
                        dt, da = H.qt_err(got["qt"], donor[k])
                        em, ea = motion_error(cks[k]["qt"], got["qt"], frames[k - 1][2], frames[k][2])
                        per[k]["builds"][name] = dict(pose_difference_m=dt, pose_difference_rad=da, motion_error_m=em, motion_error_rad=ea)
                finally:
                    api.use_library(None)
            recs += [per[k] for k in checks]
            print(f"run {hex(seed)} noise={noise} {scene} {w}x{h}: done", file=sys.stderr, flush=True)
    summary = {}
    for name, _ in columns:
        dm = np.array([r["builds"][name]["pose_difference_m"] for r in recs])
        da = np.array([r["builds"][name]["pose_difference_rad"] for r in recs])
        over = (dm > 1e-4) | (da > 1e-4)
        summary[name] = dict(checkpoints=len(recs), over_the_bar=int(over.sum()),
                             pose_difference_m=dict(median=float(np.median(dm)), p75=float(np.percentile(dm, 75)), p95=float(np.percentile(dm, 95)), max=float(dm.max())),
                             pose_difference_rad=dict(median=float(np.median(da)), p75=float(np.percentile(da, 75)), p95=float(np.percentile(da, 95)), max=float(da.max())))
    names = ["reference_rounding"] + [n for n, _ in columns]
    summary["motion_error_against_the_generating_trajectory_m"] = {
        n: dict(median=float(np.median([r["builds"][n]["motion_error_m"] for r in recs])), max=float(max(r["builds"][n]["motion_error_m"] for r in recs))) for n in names}
    over_any = [r for r in recs if any(b["pose_difference_m"] > 1e-4 or b["pose_difference_rad"] > 1e-4 for b in r["builds"].values())]
    summary["checkpoints_over_the_bar_in_any_column"] = len(over_any)
    sets = {n: {(r["seed"], r["noise"], r["scene"], tuple(r["size"]), r["frame"]) for r in recs
                if r["builds"][n]["pose_difference_m"] > 1e-4 or r["builds"][n]["pose_difference_rad"] > 1e-4} for n, _ in columns}
    summary["over_the_bar_in_all_three_columns"] = len(set.intersection(*sets.values()))
    print(json.dumps(summary, indent=1))
    with open(out_path, "w") as f:
        json.dump(dict(design=__doc__, summary=summary, over_the_bar=over_any, checkpoints=recs), f, indent=1)


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_parity_factorial.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    main(out)
