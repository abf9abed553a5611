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
