"""Scene-sharded data parallelism for the hot path.

Scenes are independent (the reference loops ``for b in range(B)`` in every stage:
``fusion_in_one.py:138``, ``enhancer.py:371``; GenComm is per-agent until fusion), so N GPUs are N
replicas, one process per GPU, each running its own shard of the scene stream with NO data-path
collective. ``torch.distributed`` is used only for the timing barrier and a max-over-ranks of the
elapsed time (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
The reference's only other distributed mechanism is DDP's gradient all-reduce for training
(``opencood/tools/train_ddp.py:121-125``) -- plain ``DistributedDataParallel`` works unchanged on
modules from this package; :class:`FlatGradSync` is the same averaging as ONE flat bucket per step.
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


class FlatGradSync:
    """Gradient averaging of data-parallel training as ONE flat bucket per step (32 MB of fp32 gradients, one RCCL
    all-reduce over xGMI = 0.4 ms link-bound; nothing to overlap with a 20 ms backward).

    ``DistributedDataParallel`` (the reference: ``train_ddp.py:121-125``) costs, per PARAMETER and step, a copy into its bucket view and a
    multiply by 1 / world on the device (520 launches for the stage-1 model's 260 tensors) plus, with ``find_unused_parameters=True``, a
    walk of the autograd graph on the host.  Here the step's gradients are concatenated (one kernel), all-reduced (one collective),
    scaled (one kernel) and copied back with a multi-tensor copy: the same sum / world on every rank.

    Semantics kept from DDP: the average over ranks of every parameter's gradient; parameters that received no gradient keep
    ``grad = None`` (the optimiser skips them, as with DDP's globally-unused parameters).  Assumed, and checked on the first
    ``check_steps`` calls with a fixed-size signature exchanged before the bucket: every rank produced gradients for the SAME
    parameters (the unused parameters of the stage-1 / stage-2 recipes are structural -- Enhancer blocks 2 and 3, frozen modules)."""

    def __init__(self, params, dist=None, check_steps: int = 2):
        self.params = [p for p in params if p.requires_grad]
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.check_steps = check_steps
        self.calls = 0
        self.bucket_bytes = 0

    def sync(self) -> None:
        """Call between ``loss.backward()`` and ``optimizer.step()``."""
        if self.dist is None:
            return
        idx = [i for i, p in enumerate(self.params) if p.grad is not None]
        grads = [self.params[i].grad for i in idx]
        if not grads:
            raise RuntimeError("FlatGradSync.sync: no parameter has a gradient (call it after backward)")
        dev, dt = grads[0].device, grads[0].dtype
        if any(g.dtype != dt or g.device != dev for g in grads):
            raise RuntimeError("FlatGradSync: gradients must share one dtype and device")
        sizes = [g.numel() for g in grads]
        if self.calls < self.check_steps and self.world > 1:
            # first steps only (a host read): every rank must be about to reduce the same parameters -- a bucket of another length would
            # hang or corrupt the collective, so this is settled on a fixed-size signature first
            sig = torch.tensor([len(idx), sum(sizes), sum((i + 1) * (i + 7) for i in idx) % (2 ** 31)], dtype=torch.int64, device=dev)
            sigs = [torch.empty_like(sig) for _ in range(self.world)]
            self.dist.all_gather(sigs, sig)
            if any(not torch.equal(s.cpu(), sig.cpu()) for s in sigs):
                raise RuntimeError("FlatGradSync: the ranks produced gradients for different sets of parameters (a parameter is unused on some "
                                   "ranks only); use torch DistributedDataParallel(find_unused_parameters=True) for such a model")
        self.calls += 1
        flat = torch.cat([g.reshape(-1) for g in grads])
        self.bucket_bytes = flat.numel() * flat.element_size()
        self.dist.all_reduce(flat)
        flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split(sizes), grads)])
