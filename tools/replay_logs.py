#!/usr/bin/env python
"""BASELINE.json configs[3]: independent 640x480 .klg sequences, one per GPU, RCCL used only to gather the throughput statistics.

    python tools/replay_logs.py a.klg b.klg ...                                                  # one GPU, one log after the other
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        tools/replay_logs.py a.klg b.klg c.klg d.klg e.klg f.klg g.klg h.klg                       # log k -> GPU k mod 8

Every log is replayed open loop through the C ABI (ef_process_frame: host frames, pinned staging, PCIe upload inside the timed
region) and leaves <log>.freiburg beside it, like the reference's front-end; rank 0 prints one JSON line with the whole-job frames/s
(all frames / the slowest rank's seconds) and the per-rank figures."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("logs", nargs="+")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--cal", type=float, nargs=4, default=[528.0, 528.0, 320.0, 240.0], metavar=("fx", "fy", "cx", "cy"))
    ap.add_argument("--close-loops", action="store_true")
    a = ap.parse_args()
    from elasticfusion_amd import multi
    rank, local_rank, world = multi.rank_info()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("replay_logs.py needs a GPU: the HIP engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        multi.init_process_group("nccl", local_rank)
    from elasticfusion_amd import api
    fx, fy, cx, cy = a.cal

    def make_engine():
        ef = api.ElasticFusion(width=a.width, height=a.height, fx=fx, fy=fy, cx=cx, cy=cy, device=local_rank,
                               **(dict(closeLoops=True, timeDelta=200) if a.close_loops else {}))
        if a.close_loops:
            ef.useBuiltinLoopSolver(True)
            ef.enableGlobalClosure(seed=0)
        return ef

    rec = multi.replay_logs(a.logs, make_engine, rank, world, a.width, a.height, on_done=lambda log, ef: ef.saveFreiburg(log + ".freiburg"))
    allstats = multi.gather_stats(rec, device="cuda")
    if rank == 0:
        agg = multi.aggregate(allstats)
        print(json.dumps({"metric": "frames/s, .klg replay", "value": round(agg["value"], 2), "unit": "frames/s", "n_gpus": world,
                          "logs": len(a.logs), "frames": int(agg["frames"]), "per_rank_fps": [round(x, 2) for x in agg["per_rank_fps"]],
                          "logs_per_rank": [int(r[2]) for r in allstats]}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
