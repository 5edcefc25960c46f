"""The N>1 path on CPU: two processes, gloo backend, world_size 2 -- scene sharding is a disjoint
cover, the timing reduction takes the slowest rank, aggregate throughput counts every rank's scenes."""
import os
import socket
import sys

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
