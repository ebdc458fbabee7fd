"""CPU test of the N>1 host path: world_size 2 over gloo (no GPU): sequence sharding + throughput reduction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from orb_slam3_amd import sharding
    seqs = sharding.sequences_for_rank(8, world, rank)
    # pretend each sequence takes (1 + rank) seconds and yields 1000 features per frame over 10 frames
    local_t = float(len(seqs)) * (1 + rank)
    local_u = float(len(seqs)) * 10 * 1000
    dist.barrier()
    t, u = sharding.reduce_throughput(local_t, local_u)
    q.put((rank, seqs, t, u))
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]
    for _, _, t, u in res:
        assert t == 8.0            # max over ranks: rank 1 needs 4 * 2 s
        assert u == 80000.0        # sum over ranks


def test_single_rank_is_identity():
    from orb_slam3_amd import sharding
    assert sharding.sequences_for_rank(3, 1, 0) == [0, 1, 2]
    assert sharding.reduce_throughput(1.5, 42.0) == (1.5, 42.0)
