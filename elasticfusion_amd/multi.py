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


def cores_of_rank(local_rank: int, ranks_on_host: int, cores=None):
    """Contiguous share of the host's cores for one rank: N ranks of one node each enqueue some twenty thousand kernel launches a second
    (a frame is ~18 launches at ~1400 frames/s) from their own host thread; left to the scheduler they migrate and share caches with the
    other ranks' frame generators.  cores: the CPUs this process may use (default: os.sched_getaffinity)."""
    cores = sorted(cores if cores is not None else os.sched_getaffinity(0))
    n = max(1, ranks_on_host)
    per = max(1, len(cores) // n)
    lo = min(local_rank * per, max(0, len(cores) - per))
    return cores[lo:lo + per]


def pin_rank_to_cores(local_rank: int, ranks_on_host: int):
    try:
        mine = cores_of_rank(local_rank, ranks_on_host)
        if mine:
            os.sched_setaffinity(0, mine)
        return mine
    except (AttributeError, OSError):   # not Linux, or a cpuset that forbids it: keep the scheduler's placement
        return None


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


def shard_logs(logs, rank: int, world: int):
    """BASELINE.json configs[3]: independent .klg sequences, one per GPU — log k goes to rank k mod world (a rank with several logs
    replays them one after the other, each in a fresh context)."""
    return [log for k, log in enumerate(logs) if k % world == rank]


class KlgFile:
    """the .klg reader of libefusion.so (include/efusion_klg.hpp: raw / zlib depth, raw / JPEG colour, the reference's frame protocol)
    through its C entry points; host-only, no GPU"""

    def __init__(self, path: str, width: int = 640, height: int = 480, all_frames: bool = False, flip_colors: bool = False):
        import ctypes as C
        here = os.path.dirname(os.path.abspath(__file__))
        self._so = C.CDLL(os.path.join(here, "libefusion.so"))
        self._so.efk_open.restype = C.c_void_p
        self._so.efk_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
        self._so.efk_last_error.restype = C.c_char_p
        for f in (self._so.efk_close, self._so.efk_num_frames, self._so.efk_has_more):
            f.argtypes = [C.c_void_p]
        self._so.efk_next.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.w, self.h = width, height
        self._h = self._so.efk_open(path.encode(), width, height, int(all_frames), int(flip_colors))
        if not self._h:
            raise IOError(self._so.efk_last_error().decode())

    def __len__(self):
        return self._so.efk_num_frames(self._h)

    def __iter__(self):
        import ctypes as C
        while self._so.efk_has_more(self._h):
            ts = C.c_int64(0)
            depth = np.zeros((self.h, self.w), np.uint16)
            rgb = np.zeros((self.h, self.w, 3), np.uint8)
            if self._so.efk_next(self._h, C.byref(ts), depth.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p)) != 1:
                raise IOError(self._so.efk_last_error().decode())
            yield ts.value, rgb, depth

    def close(self):
        if self._h:
            self._so.efk_close(self._h)
            self._h = None


def replay_logs(logs, make_engine, rank: int, world: int, width: int = 640, height: int = 480, on_done=None, chunk: int = 64):
    """rank's share of `logs` through make_engine() objects (processFrame(rgb, depth, timestamp), synchronize(), close()).  Decoding is
    host work and is kept off the clock, but a log is never held decoded as a whole (a real multi-thousand-frame log is ~1.5 MB per frame,
    times 8 ranks on one host): frames are decoded `chunk` at a time, each chunk is replayed with the clock running and the engine
    synchronised at its end.  -> [seconds, frames, logs] for gather_stats"""
    import time
    seconds, frames, done = 0.0, 0, 0
    for log in shard_logs(logs, rank, world):
        reader = KlgFile(log, width, height)
        eng = make_engine()
        it = iter(reader)
        while True:
            decoded = []
            for item in it:
                decoded.append(item)
                if len(decoded) >= chunk:
                    break
            if not decoded:
                break
            t0 = time.perf_counter()
            for ts, rgb, depth in decoded:
                eng.processFrame(rgb, depth, ts)
            eng.synchronize()
            seconds += time.perf_counter() - t0
            frames += len(decoded)
        reader.close()
        done += 1
        if on_done:
            on_done(log, eng)
        eng.close()
    return [seconds, float(frames), float(done)]
