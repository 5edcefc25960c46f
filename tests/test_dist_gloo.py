"""The N>1 path on CPU: two processes, gloo backend, world_size 2 -- scene sharding is a disjoint
cover, the timing reduction takes the slowest rank, aggregate throughput counts every rank's scenes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from gencomm_amd import dist as gd
    d = gd.init_process_group("gloo")
    dev = torch.device("cpu")
    assert gd.env_rank_world() == (rank, world, rank)
    mine = gd.shard_scenes(11, rank, world)
    # every rank learns the union through an all_gather (test-only collective)
    gathered = [None] * world
    d.all_gather_object(gathered, mine)
    elapsed = 1.0 + rank  # rank 1 is the slow replica
    thr, worst, total = gd.aggregate_throughput(len(mine), elapsed, d, dev)
    d.barrier()
    q.put((rank, mine, gathered, thr, worst, total))
    d.destroy_process_group()


def test_two_rank_scene_sharding_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mine, gathered, thr, worst, total in res:
        assert mine == list(range(rank, 11, world))
        flat = sorted(i for part in gathered for i in part)
        assert flat == list(range(11))                       # disjoint cover of the scene stream
        assert total == 11 and worst == pytest.approx(2.0)    # slowest rank defines wall time
        assert thr == pytest.approx(11 / 2.0)


def test_single_process_is_a_noop():
    from gencomm_amd import dist as gd
    assert gd.shard_scenes(5, 0, 1) == [0, 1, 2, 3, 4]
    assert gd.max_over_ranks(3.5, None, torch.device("cpu")) == 3.5
    thr, worst, total = gd.aggregate_throughput(5, 2.5, None, torch.device("cpu"))
    assert (thr, worst, total) == (2.0, 2.5, 5)
    with pytest.raises(ValueError):
        gd.shard_scenes(5, 2, 2)


# ---------------------------------------------------------------------- FlatGradSync against DistributedDataParallel
class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a, self.b = torch.nn.Linear(6, 8), torch.nn.Linear(8, 3)
        self.unused = torch.nn.Linear(4, 4)   # structurally unused on every rank, like the Enhancer's blocks 2 and 3

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _toy_model():
    torch.manual_seed(5)
    return _Toy()


def _sync_worker(rank, world, port, q, mismatch):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from gencomm_amd import dist as gd
    d = gd.init_process_group("gloo")
    g = torch.Generator().manual_seed(100 + rank)         # every rank its own shard of the data
    x, y = torch.randn(5, 6, generator=g), torch.randn(5, 3, generator=g)
    # reference: torch DDP
    ref = _toy_model()
    ddp = torch.nn.parallel.DistributedDataParallel(ref, find_unused_parameters=True)
    ((ddp(x) - y) ** 2).mean().backward()
    want = {k: (None if p.grad is None else p.grad.numpy().copy()) for k, p in ref.named_parameters()}   # numpy: plain pickles on the queue
    # one flat bucket
    m = _toy_model()
    sync = gd.FlatGradSync(m.parameters(), d)
    out = m(x)
    loss = ((out - y) ** 2).mean()
    if mismatch and rank == 1:                            # a parameter that only ONE rank uses: must be refused, not silently mis-summed
        loss = loss + m.unused(x[:, :4]).sum()
    loss.backward()
    err = None
    try:
        sync.sync()
    except RuntimeError as e:
        err = str(e)
    got = {k: (None if p.grad is None else p.grad.numpy().copy()) for k, p in m.named_parameters()}
    d.barrier()
    q.put((rank, want, got, err, sync.bucket_bytes))
    d.destroy_process_group()


def _run_sync(mismatch):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q, mismatch)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_flat_bucket_gradient_sync_equals_ddp():
    res = _run_sync(False)
    for rank, want, got, err, nbytes in res:
        assert err is None
        assert nbytes == 4 * (6 * 8 + 8 + 8 * 3 + 3)
        for k in want:
            if k.startswith("unused"):
                assert got[k] is None                       # no gradient on any rank: left alone, the optimiser skips it
                continue
            np.testing.assert_allclose(got[k], want[k], rtol=1e-6, atol=1e-7)
    a, b = res[0][2], res[1][2]
    for k in a:
        if a[k] is not None:
            assert np.array_equal(a[k], b[k])               # every rank holds the same averaged gradient


def test_flat_bucket_refuses_rank_dependent_parameter_sets():
    res = _run_sync(True)
    assert all(err is not None and "different sets of parameters" in err for _, _, _, err, _ in res)   # every rank stops, none hangs
