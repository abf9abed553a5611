"""GPU parity on the configurations bench.py actually RUNS beyond the headline:

  * BASELINE configs[2] as benched: 1280x960, a map pre-seeded with ~1 M surfels sampled on the scene (bench.preseed: ef_map_upload +
    ef_restore_state) — the same checkpoint restored into engine and oracle, three free-running frames, bit for bit (the test of rounds
    1-4, test_frames_at_1280x960_match_oracle, starts from a first-frame seed: a quarter of the surfels and none of them stable);
  * BASELINE configs[3], the per-rank path: multi.replay_logs with REAL .klg files (zlib depth) on one device — sharding, the product's
    reader, ef_process_frame, the stats record — against the oracle fed with the frames the reference's run loop delivers.
"""
import numpy as np
import pytest

import efo

pytestmark = pytest.mark.gpu


def test_config2_preseeded_map_matches_oracle():
    import bench
    from elasticfusion_amd import api, synth
    w, h, seed, n_seed = 1280, 960, 0xEF0001, 1 << 20
    sc = w / 640.0
    sq = synth.Sequence(seed, width=w, height=h)
    frames = [sq.frame(k) for k in range(4)]
    m = synth.sample_surfels(sq, n_seed)
    assert 0.9 * n_seed < len(m) < 1.2 * n_seed
    ck = dict(map=m, tick=2, qt=np.array([0, 0, 0, 1, 0, 0, 0], np.float64), rgb=frames[0][0], depth=frames[0][1])   # bench.preseed's checkpoint
    cap = max(4 * 1024 * 1024, 6 * w * h)
    ef = api.ElasticFusion(width=w, height=h, fx=528.0 * sc, fy=528.0 * sc, cx=320.0 * sc, cy=240.0 * sc, maxSurfels=cap)
    assert bench.preseed(ef, seed, w, h, n_seed, frames[0]) == len(m)      # the very call the bench leg makes
    efo.set_threads(32)
    try:
        o = efo.Fusion(width=w, height=h, fx=528.0 * sc, fy=528.0 * sc, cx=320.0 * sc, cy=240.0 * sc, maxSurfels=cap)
        o.restore(ck)
        assert ef.lastCount() == o.map_count() == len(m) and ef.getTick() == o.tick() == 2
        assert np.array_equal(ef.image("image"), o.buffer("image")) and np.array_equal(ef.image("time"), o.buffer("time"))   # the restore's predict()
        for k in range(1, 4):
            rgb, depth, _ = frames[k]
            ef.processFrame(rgb, depth, k * 33333)
            o.process_frame(rgb, depth, k * 33333)
            st = np.asarray(ef.trackingStats()[0], np.float32)
            assert np.array_equal(st.view(np.uint32), np.asarray(o.stats(), np.float32).view(np.uint32)), (k, st, o.stats())
            assert np.array_equal(ef.get_T_wc().astype(np.float32), o.pose().astype(np.float32)), k
            assert ef.lastCount() == o.map_count(), (k, ef.lastCount(), o.map_count())
        assert st[1] > 300000          # the tracker ran on the model prediction of the dense pre-seeded map (ICP inliers)
        assert np.array_equal(ef.downloadMap().view(np.uint32), o.map().view(np.uint32))
    finally:
        efo.set_threads(1)
    ef.close()


def test_replay_logs_of_real_klg_files_on_one_device(tmp_path):
    from elasticfusion_amd import api, multi, synth
    logs, want = [], {}
    for i, seed in enumerate((0xEF0001, 0xEF0002, 0xEF0003)):
        sq = synth.Sequence(seed)
        frames = [sq.frame(k) for k in range(7)]
        path = str(tmp_path / f"seq{i}.klg")
        synth.write_klg(path, frames, compress_depth=(i != 1))
        logs.append(path)
        o = efo.Fusion()
        for k, (rgb, depth, _) in enumerate(frames[:-1]):     # the reference's run loop never delivers the last frame of a log
            o.process_frame(rgb, depth, k * 33333)
        want[path] = (o.pose(), o.map_count(), o.map().copy())
    got = {}

    def on_done(log, eng):
        got[log] = (eng.get_T_wc(), eng.lastCount(), eng.downloadMap())

    total = np.zeros(3)
    for rank in range(2):    # the two ranks of a 2-GPU job, one after the other on this one device
        rec = multi.replay_logs(logs, lambda: api.ElasticFusion(), rank, 2, on_done=on_done, chunk=4)
        assert rec[2] == len(multi.shard_logs(logs, rank, 2)) and rec[1] == 6 * rec[2] and rec[0] > 0
        total += rec
    assert total[1] == 18 and total[2] == 3 and sorted(got) == sorted(logs)
    for path in logs:
        T, n, m = got[path]
        Tr, nr, mr = want[path]
        assert np.array_equal(T.astype(np.float32), Tr.astype(np.float32)) and n == nr, path
        assert np.array_equal(m.view(np.uint32), mr.view(np.uint32)), path
