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


def _sync_worker(rank, world, port, q, mode):
    """mode: "plain" one step; "union" a parameter only rank 1 uses at the FIRST step (joins the agreed layout, the others add zeros);
    "late" a parameter outside the agreed layout receives a gradient at the SECOND step on rank 1 only (refused after the collective:
    nobody hangs); "skip" a parameter of the layout has no gradient on rank 0 at the second step (zeros in, average out);
    "diverged" every rank starts from its own weights and the constructor's broadcast makes them rank 0's."""
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from gencomm_amd import dist as gd
    d = gd.init_process_group("gloo")
    g = torch.Generator().manual_seed(100 + rank)         # every rank its own shard of the data
    x, y = torch.randn(5, 6, generator=g), torch.randn(5, 3, generator=g)
    # reference: torch DDP (mode "plain"), else the average over ranks of the local gradients with zeros where a rank has none
    if mode == "plain":
        ref = _toy_model()
        ddp = torch.nn.parallel.DistributedDataParallel(ref, find_unused_parameters=True)
        ((ddp(x) - y) ** 2).mean().backward()
        want = {k: (None if p.grad is None else p.grad.numpy().copy()) for k, p in ref.named_parameters()}   # numpy: plain pickles on the queue
    else:
        ref = _toy_model()
        ref_loss = ((ref(x) - y) ** 2).mean()
        if mode == "union" and rank == 1:
            ref_loss = ref_loss + ref.unused(x[:, :4]).sum()
        ref_loss.backward()
        local = {k: (None if p.grad is None else p.grad.numpy().copy()) for k, p in ref.named_parameters()}
        every = [None] * world
        d.all_gather_object(every, local)
        want = {}
        for k, p in ref.named_parameters():
            have = [e[k] for e in every if e[k] is not None]
            want[k] = None if not have else sum(have) / world
    # one flat bucket
    if mode == "diverged":
        torch.manual_seed(50 + rank)
        m = _Toy()
    else:
        m = _toy_model()
    sync = gd.FlatGradSync(m.parameters(), d, module=m)
    start = {k: p.detach().numpy().copy() for k, p in m.named_parameters()}
    err = None
    steps = 2 if mode in ("late", "skip") else 1
    for step in range(steps):
        for p_ in m.parameters():
            p_.grad = None
        out = m(x)
        loss = ((out - y) ** 2).mean()
        if mode == "union" and rank == 1:
            loss = loss + m.unused(x[:, :4]).sum()
        if mode == "late" and rank == 1 and step == 1:
            loss = loss + m.unused(x[:, :4]).sum()
        if mode == "skip" and rank == 0 and step == 1:
            loss = ((torch.tanh(m.a(x))[:, :3] - y) ** 2).mean()      # m.b unused on rank 0 this step
        loss.backward()
        try:
            sync.sync()
        except RuntimeError as e:
            err = str(e)
    got = {k: (None if p.grad is None else p.grad.numpy().copy()) for k, p in m.named_parameters()}
    d.barrier()
    q.put((rank, want, got, err, sync.bucket_bytes, start))
    d.destroy_process_group()


def _run_sync(mode, world=2):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


def _check_equals_ddp(res, unused_is_none=True):
    for rank, want, got, err, nbytes, _ in res:
        assert err is None
        for k in want:
            if k.startswith("unused") and unused_is_none:
                assert got[k] is None                       # no gradient on any rank: left alone, the optimiser skips it
                continue
            np.testing.assert_allclose(got[k], want[k], rtol=1e-6, atol=1e-7)
    a = res[0][2]
    for r in res[1:]:
        for k in a:
            if a[k] is not None:
                assert np.array_equal(a[k], r[2][k])        # every rank holds the same averaged gradient


def test_flat_bucket_gradient_sync_equals_ddp():
    res = _run_sync("plain")
    assert all(r[4] == 4 * (6 * 8 + 8 + 8 * 3 + 3) for r in res)
    _check_equals_ddp(res)


def test_flat_bucket_at_eight_ranks_equals_ddp():
    """world size 8 (the node the benchmark targets, reference multi_gpu_utils.py:16-38): same average as DistributedDataParallel"""
    _check_equals_ddp(_run_sync("plain", world=8))


def test_flat_bucket_layout_is_the_union_over_ranks():
    """a parameter that only ONE rank uses at the first step joins the layout: the other ranks contribute zeros and receive the
    average, exactly as DistributedDataParallel(find_unused_parameters=True) does -- no rank reduces a bucket of another length"""
    res = _run_sync("union")
    assert all(r[4] == 4 * (6 * 8 + 8 + 8 * 3 + 3 + 4 * 4 + 4) for r in res)
    _check_equals_ddp(res, unused_is_none=False)


def test_flat_bucket_parameter_without_gradient_on_one_rank_gets_the_average():
    res = _run_sync("skip")
    assert all(r[3] is None for r in res)
    for k in ("b.weight", "b.bias"):
        assert res[0][2][k] is not None and np.array_equal(res[0][2][k], res[1][2][k])   # rank 0 added zeros and holds rank 1's half


def test_flat_bucket_refuses_late_parameters_without_hanging():
    res = _run_sync("late")
    assert res[0][3] is None and res[1][3] is not None and "outside the layout" in res[1][3]   # the collective completed on both ranks
    for k in ("a.weight", "b.weight"):
        assert np.array_equal(res[0][2][k], res[1][2][k])


def test_flat_bucket_broadcasts_rank0_state():
    res = _run_sync("diverged")
    for k in res[0][5]:
        assert np.array_equal(res[0][5][k], res[1][5][k])
