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

    Semantics kept from DDP:
    * construction broadcasts rank 0's parameters (and, with ``module=``, its buffers) so that every rank starts from one state
      (DDP's ``_sync_module_states``); ``broadcast=False`` skips it for callers that seeded identically;
    * the bucket LAYOUT is fixed at the first ``sync()``: the union over ranks of the parameters that received a gradient (one
      all-reduce(MAX) of a per-parameter mask, one host read, first step only).  From then on every rank reduces exactly that
      layout -- a parameter of the set without a local gradient on some step contributes zeros and receives the average (DDP's
      locally-unused parameter), so the collective's length can never differ between ranks;
    * parameters outside the set keep ``grad = None`` (the optimiser skips them, as with DDP's globally-unused parameters: Enhancer
      blocks 2 and 3, frozen modules).  A gradient that appears LATER on a parameter outside the set is an error (raised after the
      collective, so the other ranks are not left inside it): such a model needs DistributedDataParallel(find_unused_parameters=True)."""

    def __init__(self, params, dist=None, check_steps: int = 2, module: torch.nn.Module = None, broadcast: bool = True):
        self.params = [p for p in params if p.requires_grad]
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.check_steps = check_steps   # kept for callers of the round-4 signature; the layout agreement replaced the per-step check
        self.calls = 0
        self.bucket_bytes = 0
        self.used = None                 # indices into self.params of the agreed layout
        if broadcast and dist is not None and self.world > 1:
            tensors = [p.data for p in self.params]
            if module is not None:
                tensors += [b.data for b in module.buffers() if b is not None]
            self._broadcast_(tensors)

    def _broadcast_(self, tensors) -> None:
        """rank 0's values into every rank, one broadcast per (dtype, device) group"""
        groups = {}
        for t in tensors:
            groups.setdefault((t.dtype, t.device), []).append(t)
        for (dt, dev), ts in groups.items():
            if dt == torch.bool:
                flat = torch.cat([t.reshape(-1).to(torch.uint8) for t in ts])
            else:
                flat = torch.cat([t.reshape(-1) for t in ts])
            self.dist.broadcast(flat, src=0)
            outs = flat.split([t.numel() for t in ts])
            with torch.no_grad():
                for t, o in zip(ts, outs):
                    t.copy_(o.view_as(t).to(dt))

    def _agree_layout(self, dev) -> None:
        mask = torch.tensor([1 if p.grad is not None else 0 for p in self.params], dtype=torch.int32, device=dev)
        if self.world > 1:
            self.dist.all_reduce(mask, op=self.dist.ReduceOp.MAX)
        self.used = [i for i, m in enumerate(mask.cpu().tolist()) if m]
        if not self.used:
            raise RuntimeError("FlatGradSync.sync: no parameter has a gradient on any rank (call it after backward)")
        self.sizes = [self.params[i].numel() for i in self.used]
        self.used_set = set(self.used)

    def sync(self) -> None:
        """Call between ``loss.backward()`` and ``optimizer.step()``."""
        if self.dist is None:
            return
        first = next((p.grad for p in self.params if p.grad is not None), None)
        ref = first if first is not None else self.params[0]
        dev, dt = ref.device, ref.dtype
        if self.used is None:
            self._agree_layout(dev)
        late = [i for i, p in enumerate(self.params) if p.grad is not None and i not in self.used_set]
        grads, missing = [], []
        for i in self.used:
            p = self.params[i]
            if p.grad is None:               # used elsewhere (or earlier): zeros in, the average out
                p.grad = torch.zeros_like(p)
                missing.append(i)
            grads.append(p.grad)
        if any(g.dtype != dt or g.device != dev for g in grads):
            raise RuntimeError("FlatGradSync: gradients must share one dtype and device")
        self.calls += 1
        flat = torch.cat([g.reshape(-1) for g in grads])
        self.bucket_bytes = flat.numel() * flat.element_size()
        self.dist.all_reduce(flat)
        flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split(self.sizes), grads)])
        if late:
            raise RuntimeError(f"FlatGradSync: {len(late)} parameter(s) outside the layout agreed at the first step received a gradient "
                               "(conditionally used parameters); use torch DistributedDataParallel(find_unused_parameters=True) for such a model")
