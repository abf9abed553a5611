"""N > 1 path on CPU: two processes over gloo run the same rank-sharding + stats-gather code bench.py uses under
torchrun with RCCL (elasticfusion_amd/multi.py).  "Replicas only": no data-path collective exists to test."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    from elasticfusion_amd import multi
    rank, local_rank, world = multi.rank_info()
    dist = multi.init_process_group("gloo", local_rank)
    seed = multi.sequence_seed(rank)
    # rank r pretends to have replayed 100 frames of ITS sequence in (1 + r) seconds
    allstats = multi.gather_stats([1.0 + rank, 100.0, 0.001 * rank, float(seed)])
    dist.barrier()
    agg = multi.aggregate(allstats)
    if rank == 0:
        print(json.dumps({"world": world, "seeds": [int(x) for x in allstats[:, 3]], **agg}))
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_gloo_stats_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["world"] == 2
    assert out["seeds"] == [0xEF0001, 0xEF0002]            # sequence k -> rank k
    assert out["frames"] == 200.0 and out["t_max"] == 2.0   # whole-job frames / slowest rank
    assert abs(out["value"] - 100.0) < 1e-9
    assert out["per_rank_fps"] == [100.0, 50.0]


def test_single_process_path_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from elasticfusion_amd import multi
    a = multi.gather_stats([2.0, 50.0, 0.0, 1.0])
    assert a.shape == (1, 4)
    assert multi.aggregate(a)["value"] == 25.0


KLG_WORKER = textwrap.dedent("""
    import json, os, sys, zlib
    sys.path.insert(0, %r)
    import numpy as np
    from elasticfusion_amd import multi
    rank, local_rank, world = multi.rank_info()
    dist = multi.init_process_group("gloo", local_rank)
    logs = sys.argv[1:]

    class Engine:                                  # stands where api.ElasticFusion stands on the GPU box
        def __init__(self): self.crc, self.n, self.ts = 0, 0, []
        def processFrame(self, rgb, depth, ts):
            self.crc = zlib.crc32(depth.tobytes(), zlib.crc32(rgb.tobytes(), self.crc)); self.n += 1; self.ts.append(ts)
        def synchronize(self): pass
        def close(self): pass

    seen = {}
    rec = multi.replay_logs(logs, Engine, rank, world, on_done=lambda log, e: seen.update({os.path.basename(log): (e.n, e.crc, e.ts)}))
    allstats = multi.gather_stats(rec)
    mine = [None] * world
    dist.all_gather_object(mine, seen)
    if rank == 0:
        agg = multi.aggregate(allstats)
        print(json.dumps({"frames": agg["frames"], "logs_per_rank": [int(r[2]) for r in allstats], "seen": mine}))
    dist.destroy_process_group()
""") % ROOT


def test_klg_logs_are_sharded_one_per_rank(tmp_path):
    """BASELINE configs[3] on CPU: three .klg files (raw, zlib depth, different lengths), two gloo ranks — log k goes to rank k mod 2, each
    rank decodes its logs through the product's reader (libefusion.so, host only) frame by frame in the reference's protocol (the last
    frame of a log is never delivered), and the only collective is the gather of {seconds, frames, logs}"""
    import json
    import zlib
    import numpy as np
    sys.path.insert(0, ROOT)
    from elasticfusion_amd import build, synth
    build.build()
    seq = synth.Sequence(0xEF0001)
    lens = [4, 3, 5]
    logs, want = [], {}
    for i, n in enumerate(lens):
        frames = [seq.frame(3 * i + k) for k in range(n)]
        path = str(tmp_path / ("log%d.klg" % i))
        synth.write_klg(path, frames, compress_depth=(i == 1))
        logs.append(path)
        crc = 0
        for rgb, depth, _ in frames[:-1]:                       # RawLogReader::hasMore: currentFrame + 1 < numFrames
            crc = zlib.crc32(depth.tobytes(), zlib.crc32(rgb.tobytes(), crc))
        want[os.path.basename(path)] = (n - 1, crc)
    script = tmp_path / "klg_worker.py"
    script.write_text(KLG_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", str(script)] + logs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["logs_per_rank"] == [2, 1] and out["frames"] == sum(n - 1 for n in lens)
    assert sorted(out["seen"][0]) == ["log0.klg", "log2.klg"] and sorted(out["seen"][1]) == ["log1.klg"]
    for per_rank in out["seen"]:
        for name, (n, crc, ts) in per_rank.items():
            assert (n, crc) == want[name], name
            assert ts == sorted(ts) and len(ts) == n
