"""Steady-state parity at the reference's DEFAULT thresholds (confidence 10, open loop), 640x480, 130 free-running frames.

With confidence = 10 and a quarter of the pixels fused per frame (quirk Q12) a surfel needs some twenty observations to
become stable, so only a long run reaches the regime the headline metric is measured in: stable surfels, a tracker fed
by the MODEL prediction (denseEnough, ElasticFusion.cpp:256-268,304-305), combinedPredict with real overdraw, clean()
removing stale unstable surfels (copy_unstable.vert:114-120).  Three comparisons:

  1. the shipped default (reference rounding: no fused multiply-add, the reference's summation order) vs the oracle (== the
     reference's own sources compiled without contraction), bit for bit, frame after frame, with and without sensor noise;
  2. the opt-in fast build vs its oracle, bit for bit: tests/test_gpu_fast_build.py::test_steady_state_equals_the_fast_oracle;
  3. default vs fast build, both free-running: the divergence two legitimate roundings of the same arithmetic
     accumulate over 130 frames, held against the north_star bars (pose 1e-4 m / 1e-4 rad, surfels 1e-5 relative).
"""
import json
import os

import numpy as np
import pytest

import efo

pytestmark = pytest.mark.gpu
N = 130
CHECKPOINTS = (30, 60, 100, N - 1)
SEQS = {"clean": dict(seed=0xEF0002, noise=False), "noisy": dict(seed=0xEF0006, noise=True)}
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _frame_job(args):
    seed, k = args
    from elasticfusion_amd import synth
    s = _frame_job.cache.get(seed)
    if s is None:
        s = _frame_job.cache[seed] = synth.Sequence(seed)
    return s.frame(k)


_frame_job.cache = {}


def make_sequences(names=("clean", "noisy")):
    """{name: [(rgb, depth, T_wc)] * N} (spawned workers: the parent may already hold a HIP runtime)"""
    import multiprocessing as mp
    from elasticfusion_amd import synth
    out = {}
    ctx = mp.get_context("spawn")
    with ctx.Pool(max(1, min(16, (os.cpu_count() or 2) - 1))) as pool:
        pending = pool.map_async(_frame_job, [(SEQS["clean"]["seed"], k) for k in range(N)], chunksize=4)
        if "noisy" in names:
            s = synth.Sequence(SEQS["noisy"]["seed"], noise=True)   # the noise generator is one stream: these frames are made in order
            out["noisy"] = [s.frame(k) for k in range(N)]
        out["clean"] = pending.get()
    return out


@pytest.fixture(scope="module")
def sequences():
    """generated once per module"""
    return make_sequences()


def dense_enough(image_rgba):
    """ElasticFusion::denseEnough (ElasticFusion.cpp:256-268) on the predicted image: Resize::image samples texel (20a+10, 20b+10)"""
    s = image_rgba[10::20, 10::20, :3]
    return float(((s[..., 0] > 0) & (s[..., 1] > 0) & (s[..., 2] > 0)).mean()) > 0.75


def run_oracle(frames):
    efo.set_threads(min(os.cpu_count() or 1, 32))   # results do not depend on the thread count (oracle/efo_common.h)
    o = efo.Fusion()
    rec = dict(stats=[], pose=[], count=[], maps={})
    for k, (rgb, depth, _) in enumerate(frames):
        o.process_frame(rgb, depth, k * 33333)
        rec["stats"].append(o.stats().copy())
        rec["pose"].append(o.pose().copy())
        rec["count"].append(o.map_count())
        if k in CHECKPOINTS:
            rec["maps"][k] = o.map()
    del o
    efo.set_threads(1)
    return rec


def run_hip(api, frames, every_map=False):
    ef = api.ElasticFusion()
    rec = dict(stats=[], pose=[], count=[], maps={}, dense=[], stable=[], removed_old=[], stale_left=[], stale_candidates=[])
    prev = None
    thr = ef.getConfidenceThreshold()
    for k, (rgb, depth, _) in enumerate(frames):
        ef.processFrame(rgb, depth, k * 33333)
        st, _, _ = ef.trackingStats()
        rec["stats"].append(np.asarray(st, np.float32))
        rec["pose"].append(ef.get_T_wc())
        rec["count"].append(ef.lastCount())
        if every_map or k in CHECKPOINTS:
            m = ef.downloadMap()
            if k in CHECKPOINTS:
                rec["maps"][k] = m
            if every_map:
                t = k + 1                                       # the tick this frame was fused at
                unstable = m[:, 3] < thr
                rec["stable"].append(int((~unstable).sum()))
                rec["stale_left"].append(int((unstable & (t - m[:, 7] > 20)).sum()))   # copy_unstable.vert:120 must have removed these
                if prev is not None:
                    born = int((m[:, 6] == t).sum())
                    rec["removed_old"].append(len(prev) + born - len(m))
                    rec["stale_candidates"].append(int(((prev[:, 3] < thr) & (t - prev[:, 7] > 20)).sum()))
                prev = m
                rec["dense"].append(dense_enough(ef.image("image")))
    ef.close()
    return rec


def assert_same_run(h, o, tag):
    for k in range(N):
        a, b = h["stats"][k], np.asarray(o["stats"][k], np.float32)
        if k > 0:
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (tag, k, a, b)
        assert np.array_equal(h["pose"][k].astype(np.float32), o["pose"][k].astype(np.float32)), (tag, k)
        assert np.abs(h["pose"][k] - o["pose"][k]).max() <= 1e-15, (tag, k)
        assert h["count"][k] == o["count"][k], (tag, k, h["count"][k], o["count"][k])
    for k in CHECKPOINTS:
        assert np.array_equal(h["maps"][k].view(np.uint32), o["maps"][k].view(np.uint32)), (tag, k)


@pytest.mark.parametrize("name", ["clean", "noisy"])
def test_default_config_reaches_steady_state_and_matches_oracle(sequences, name):
    from elasticfusion_amd import api
    frames = sequences[name]
    h = run_hip(api, frames, every_map=True)
    # the regimes the short tests never reach, asserted on the way
    assert h["stable"][10] == 0 and h["stable"][-1] > 50000, (h["stable"][10], h["stable"][-1])          # stable surfels appear
    first_stable = next(k for k, v in enumerate(h["stable"]) if v > 0)
    assert 15 <= first_stable <= 60, first_stable
    assert not h["dense"][5] and h["dense"][-1], (h["dense"][5], h["dense"][-1])                            # fill-in maps -> model prediction
    flip = next(k for k, v in enumerate(h["dense"]) if v)
    assert first_stable <= flip <= N - 15, (first_stable, flip)   # at least 15 frames with the tracker fed by the model prediction
    assert max(h["stale_left"]) == 0                                                                     # nothing stale survives a clean()
    assert sum(c > 0 for c in h["stale_candidates"]) > 10 and sum(h["removed_old"]) > 1000, (sum(h["stale_candidates"]), sum(h["removed_old"]))
    o = run_oracle(frames)
    assert_same_run(h, o, name)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, f"steady_{name}.json"), "w") as f:
        json.dump(dict(frames=N, first_stable_frame=first_stable, dense_enough_from_frame=flip, stable_surfels_end=h["stable"][-1],
                       surfels_end=h["count"][-1], old_surfels_removed=int(sum(h["removed_old"])),
                       frames_with_stale_candidates=int(sum(c > 0 for c in h["stale_candidates"]))), f)


def pose_err(T, Tr):
    dt = float(np.linalg.norm(T[:3, 3] - Tr[:3, 3]))
    dR = T[:3, :3].T @ Tr[:3, :3]
    return dt, float(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))


@pytest.mark.fastbuild
def test_fma_placement_divergence_free_running(sequences):
    """The shipped default (reference rounding) vs the opt-in fast build (fused multiply-adds + fast order), both free-running on the
    same frames.  nvcc's actual FMA choices cannot be observed here; the two builds bracket them (nothing fused vs everything the
    specification fuses), so their divergence says how far a real nvcc build of the reference can sit from either.

    MEASURED (MI355X, round 2, gpurun_out/fma_divergence.json): the two roundings part at the second frame and sit
    1-3 mm / 1-2 mrad apart after 130 frames — the tracker is a feedback loop (pose -> association -> map -> pose) that amplifies a
    1-ulp difference, so BASELINE.json's bars (pose 1e-4 m / 1e-4 rad, surfels 1e-5 relative) hold between two implementations
    only when their rounding is IDENTICAL (which is what the bit-exact tests above establish against the oracle and, in the no-FMA
    build, against the reference's own arithmetic); they do not hold between two legitimate roundings of the same arithmetic.
    Both builds stay equally close to the generating trajectory.  The assertions below pin the measured envelope."""
    from scipy.spatial import cKDTree
    from elasticfusion_amd import api, build
    frames = sequences["clean"]
    b = run_hip(api, frames)                    # b: the shipped default = reference rounding
    api.use_library(build.FAST_LIB)
    try:
        a = run_hip(api, frames)                # a: the opt-in fast build
    finally:
        api.use_library(None)
    errs = [pose_err(a["pose"][k], b["pose"][k]) for k in range(N)]
    max_dt, max_da = max(e[0] for e in errs), max(e[1] for e in errs)
    gt = [frames[k][2] for k in range(N)]
    err_a = [pose_err(a["pose"][k], gt[k])[0] for k in range(N)]
    err_b = [pose_err(b["pose"][k], gt[k])[0] for k in range(N)]
    ma, mb = a["maps"][N - 1], b["maps"][N - 1]
    # surfels are matched by position (association decisions differ, so the two maps need not have the same length)
    d, idx = cKDTree(mb[:, :3].astype(np.float64)).query(ma[:, :3].astype(np.float64))
    scale = np.linalg.norm(ma[:, :3], axis=1)
    rec = dict(frames=N, max_pose_divergence_m=max_dt, max_pose_divergence_rad=max_da, final_pose_divergence_m=errs[-1][0],
               surfels_fast_build=int(len(ma)), surfels_reference_rounding=int(len(mb)),
               fraction_position_within_1e5_relative=float((d <= 1e-5 * scale).mean()),
               fraction_position_within_5mm=float((d <= 5e-3).mean()), median_surfel_distance_m=float(np.median(d)),
               max_err_vs_generating_traj_fast_build_m=max(err_a), max_err_vs_generating_traj_reference_rounding_m=max(err_b),
               north_star_pose_bar_met=bool(max_dt <= 1e-4 and max_da <= 1e-4),
               identical_trajectory_frames=int(sum(np.array_equal(a["pose"][k], b["pose"][k]) for k in range(N))))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "fma_divergence.json"), "w") as f:
        json.dump(rec, f)
    print("FMA divergence:", rec)
    assert max_dt <= 1e-2 and max_da <= 1e-2, rec                       # millimetres, not centimetres
    assert abs(len(ma) - len(mb)) <= 0.01 * len(ma), rec
    assert rec["fraction_position_within_5mm"] >= 0.99, rec
    assert abs(max(err_a) - max(err_b)) <= 5e-3, rec                   # neither rounding tracks the generating trajectory better
