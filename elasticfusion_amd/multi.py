"""Multi-GPU driver pieces shared by bench.py and the CPU (gloo) tests.

The per-frame path does not shard inside one sequence (an ICP iteration would need a 29-float all-reduce
57 times per frame) — "replicas only": rank r owns GPU r and replays its own independent sequence.  The one collective
is an all-gather of a 4-double stats record at the end of the run (RCCL over xGMI on the GPU box, gloo in the tests).
"""
from __future__ import annotations

import os

import numpy as np

BASE_SEED = 0xEF0001   # sequence seeds 0xEF0001 .. 0xEF0008


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def sequence_seed(rank: int) -> int:
    """Sequence k goes to GPU k."""
    return BASE_SEED + rank


def init_process_group(backend: str, local_rank: int):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # the host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails (hipIpcGetMemHandle)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist


def gather_stats(record, device=None):
    """all_gather of one small float64 record per rank -> (world, len(record)) numpy array on every rank."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(record), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        return torch.stack(out).cpu().numpy()
    return t.cpu().numpy()[None]


def aggregate(allstats: np.ndarray):
    """allstats rows = [seconds, frames, ...]: whole-job throughput = all frames / slowest rank's time."""
    t_max = float(allstats[:, 0].max())
    frames = float(allstats[:, 1].sum())
    return {"value": frames / t_max, "t_max": t_max, "frames": frames,
            "per_rank_fps": [float(r[1] / r[0]) for r in allstats]}
