"""Scene-sharded data parallelism for the hot path.

Scenes are independent (the reference loops ``for b in range(B)`` in every stage:
``fusion_in_one.py:138``, ``enhancer.py:371``; GenComm is per-agent until fusion), so N GPUs are N
replicas, one process per GPU, each running its own shard of the scene stream with NO data-path
collective. ``torch.distributed`` is used only for the timing barrier and a max-over-ranks of the
elapsed time (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
The reference's only other distributed mechanism is DDP's gradient all-reduce for training
(``opencood/tools/train_ddp.py:121-125``) -- plain ``DistributedDataParallel`` works unchanged on
modules from this package once backward exists; nothing here re-implements it.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun / torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend: str, device: Optional[torch.device] = None, force: bool = False):
    """Initialise torch.distributed from the environment; returns the module (or None when
    WORLD_SIZE == 1 and not ``force`` -- the training leg creates a one-rank group so that DDP and the RCCL
    communicator run the same code at every N). 127.0.0.1 is the default rendezvous address (container hostnames
    may not resolve)."""
    rank, world, _ = env_rank_world()
    if world == 1 and not force:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        from .launch import free_port
        os.environ["MASTER_PORT"] = str(free_port()) if world == 1 else "29500"
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


def shard_scenes(num_scenes: int, rank: int, world: int) -> List[int]:
    """Round-robin shard of a stream of scene indices: rank r gets r, r+world, r+2*world, ...
    Every scene is processed by exactly one rank; shards differ in size by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} not in [0, {world})")
    return list(range(rank, num_scenes, world))


def max_over_ranks(value: float, dist, device: torch.device) -> float:
    """Max of a host scalar over all ranks (the slowest replica defines the job's wall time)."""
    if dist is None:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist, device: torch.device) -> float:
    if dist is None:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def aggregate_throughput(scenes_this_rank: int, elapsed_this_rank: float, dist, device: torch.device) -> Tuple[float, float, int]:
    """Whole-job scenes/sec = (scenes processed by ALL ranks) / (max elapsed over ranks).
    Returns (throughput, max_elapsed, total_scenes)."""
    total = int(round(sum_over_ranks(float(scenes_this_rank), dist, device)))
    worst = max_over_ranks(elapsed_this_rank, dist, device)
    return total / worst, worst, total
