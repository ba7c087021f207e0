"""Multi-GPU path on CPU: the gap list is sharded with no data-path collective
(gloo, world_size 2, covers bench.py's rendezvous + max/sum reduction and the
static-chunk + stealing assignment)."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gap2seq_amd import shard  # noqa: E402


def test_chunks_partition_the_list():
    b = shard.chunk_bounds(1000, 64)
    assert b[0] == (0, 64) and b[-1] == (960, 1000)
    assert sum(e - s for s, e in b) == 1000


def test_assignment_is_a_partition_and_balanced():
    costs = [100] * 7 + [1000, 10, 10, 5000, 3, 3, 3, 3, 3]
    for world in (1, 2, 3, 4, 8):
        owner = shard.assign_chunks(costs, world)
        flat = sorted(c for o in owner for c in o)
        assert flat == list(range(len(costs)))
        load = [sum(costs[c] for c in o) for o in owner]
        # never worse than plain round robin
        rr = [sum(costs[c] for c in range(r, len(costs), world)) for r in range(world)]
        assert max(load) <= max(rr)


def test_shards_cover_all_gaps_for_every_world_size():
    n = 10000
    costs = [shard.gap_cost(200 + (i * 37) % 800, 500, 10, 10) for i in range(n)]
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            idx = shard.shard_for_rank(n, costs, r, world, chunk=64)
            assert idx == sorted(idx)
            seen.extend(idx)
        assert sorted(seen) == list(range(n))


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000
    costs = [shard.gap_cost(200 + (i * 37) % 800, 500, 10, 10) for i in range(n)]
    mine = shard.shard_for_rank(n, costs, rank, world, chunk=16)
    dist.barrier()
    secs, units = shard.reduce_timing(0.5 + rank, float(len(mine)), dist)
    q.put((rank, secs, units, len(mine)))
    dist.destroy_process_group()


def test_gloo_two_ranks_no_data_collective():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    assert out[0][1] == out[1][1] == 1.5          # max over ranks
    assert out[0][2] == out[1][2] == 1000.0       # whole-job units
    assert out[0][3] + out[1][3] == 1000
