"""CPU: bench.py's OWN multi-rank control flow (tests/test_multi_gloo.py covers elasticfusion_amd/multi.py, not this file's `world > 1`
branches): the driver's command line for N = 2 — `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2` —
with the gloo backend and a stand-in engine (--stand-in-engine: counts frames, sleeps 2 ms per frame): both ranks pin themselves to their
share of the host's cores, meet at the barriers around the timed region, gather the 32-byte stats record once; rank 0 prints ONE JSON
line with the whole-job aggregate (all frames / the slowest rank's time), rank 1 leaves silently; both exit 0."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_gloo_stand_in_engine():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--preroll", "2", "--width", "64",
           "--height", "48", "--stand-in-engine", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env, timeout=280)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 alone prints the line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 3 and d["scaling"] == "weak" and d["data"] == "stand-in"
    assert d["config"]["sequences"] == 2 and d["config"]["rccl_world_size"] == 2 and len(d["config"]["per_rank_fps"]) == 2
    # whole-job aggregate: 2 x 20 frames over the slowest rank's time; 2 ms per stand-in frame bounds it from above
    assert abs(d["value"] - 40.0 / (d["ms_per_step"] * 20 / 1e3)) <= 0.02 * d["value"]
    assert d["value"] <= 2 * 500.0 * 1.01 and min(d["config"]["per_rank_fps"]) <= 500.0 * 1.01
    assert d["roofline"] is None and d["vs_baseline"] is None


def test_bench_eight_ranks_gloo_the_real_width():
    """the width the driver's scaling run uses, on this 8-core box — eight ranks, each forking the REAL frame generator
    (tiny frames), each pinned to its own core share (multi.pin_rank_to_cores), the stand-in engine, one collective: ONE JSON line from rank 0,
    the other seven silent, every exit code 0, the whole-job aggregate over eight sequences, and the per-rank seeds 0xEF0001 .. 0xEF0008 as
    gathered by the collective itself (sequence k goes to GPU k)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "10", "--warmup", "2", "--preroll", "2", "--width", "64",
           "--height", "48", "--stand-in-engine", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env, timeout=560)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 10 and d["scaling"] == "weak" and d["data"] == "stand-in"
    assert d["config"]["sequences"] == 8 and d["config"]["rccl_world_size"] == 8 and len(d["config"]["per_rank_fps"]) == 8
    assert d["config"]["sequence_seeds"] == [hex(0xEF0001 + k) for k in range(8)], d["config"]["sequence_seeds"]
    assert abs(d["value"] - 80.0 / (d["ms_per_step"] * 10 / 1e3)) <= 0.02 * d["value"]
    assert d["value"] <= 8 * 500.0 * 1.01


def test_ranks_get_disjoint_contiguous_core_shares():
    from elasticfusion_amd import multi
    cores = list(range(3, 35))            # a 32-core cpuset that does not start at 0
    shares = [multi.cores_of_rank(r, 8, cores) for r in range(8)]
    assert all(len(s) == 4 for s in shares) and sorted(sum(shares, [])) == cores
    assert all(s == list(range(s[0], s[0] + 4)) for s in shares)
    assert multi.cores_of_rank(0, 1, cores) == cores
    assert len(multi.cores_of_rank(5, 8, [0, 1, 2])) == 1      # fewer cores than ranks: one each, shared
