"""The N>1 path on CPU: one gap list spread over the sessions of the dispatcher with no
data-path collective.  Covered here without a GPU: the choice of the group size, the shared
group counter g2s_team_fill's sessions pull from (C++ test hook, host threads in place of
sessions), and bench.py's launch protocol under torch.distributed.run with 2 ranks (gloo):
rendezvous on 127.0.0.1, barriers around the timed region, max/sum reduction, one JSON line."""
import json
import os
import socket
import subprocess

import pytest
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gap2seq_amd import shard  # noqa: E402


def test_group_size_and_bounds_partition_the_list():
    for n, sessions in ((10000, 1), (10000, 2), (10000, 4), (10000, 8), (10000, 16), (500, 8), (7, 3)):
        g = shard.group_size(n, sessions)
        b = shard.group_bounds(n, g)
        assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        assert len(b) <= sessions  # one launch per session at most: a launch costs its slowest gap
    assert shard.group_size(10000, 8) == 1250 and shard.group_size(500, 8) == 256


def test_group_queue_hands_every_gap_out_once(product):
    for workers, n, group in ((1, 1000, 64), (2, 10000, 1250), (8, 10000, 1250), (8, 10000, 100), (3, 17, 5), (4, 5, 100)):
        owner = product.test_group_queue(workers, n, group)
        assert len(owner) == n and all(0 <= o < workers for o in owner)
        for b, e in shard.group_bounds(n, group):  # a group goes to one worker as a whole
            assert len(set(owner[b:e])) == 1
    assert product.test_group_queue(4, 0, 16) == []


def test_a_slow_sessions_groups_are_taken_by_the_others(product):
    """The dispatcher's counter hands a group to whoever asks (the reference's threads pull scaffolds from a
    shared iterator the same way, Gap2Seq.cpp:313-323): with more groups than workers, a worker that needs 3 ms
    more per group than the others (a busy device) ends up with a handful of the 80 groups, the other three share
    the rest, and every gap is still handed out exactly once."""
    n, group, workers = 4000, 50, 4
    assert shard.group_size(n, workers, min_group=1, per_session=20) == group
    owner = product.test_group_queue_slow(workers, n, group, 0, 3000)
    per_worker = [sum(1 for b, _ in shard.group_bounds(n, group) if owner[b] == w) for w in range(workers)]
    assert sum(per_worker) == 80 and len(owner) == n and all(0 <= o < workers for o in owner)
    assert per_worker[0] <= 4, per_worker                      # (a static split would have left it 20)
    assert min(per_worker[1:]) >= 15, per_worker               # the others took what it did not get to
    even = product.test_group_queue_slow(workers, n, group, -1, 0)
    counts = [sum(1 for b, _ in shard.group_bounds(n, group) if even[b] == w) for w in range(workers)]
    assert sum(counts) == 80 and min(counts) >= 8, counts      # nobody slow: everybody gets a fair share


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_launch_protocol_two_ranks_gloo():
    """What the driver does for N>1: torch.distributed.run starts one rank per GPU; rank 0's
    process drives the N devices, the other ranks only join the barriers; exactly one JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--dry-run"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["dry_run"] and out["n_gpus"] == 2 and out["ranks_under_torchrun"] == 2
    assert out["units"] == 1.0 and out["elapsed_max_over_ranks_s"] >= 0.01
    assert (out["group"], out["groups"]) == (5000, 2)


def test_bench_refuses_more_gpus_than_there_are():
    """`--gpus N` is never silently reduced: without N usable devices the run fails."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode != 0
    assert "device" in (res.stdout + res.stderr)


# ---- round 6: one rank per GPU (a launcher that pins a device per rank): the shares of a list placed in the one
# rand() stream by two all-gathers of host scalars (gap2seq_amd/shard.py: place_shares, DistComm, fill_share)
def test_place_shares_composes_the_share_functions():
    from gap2seq_amd import shard
    assert shard.share_bounds(10, 3) == [(0, 3), (3, 6), (6, 10)]
    assert shard.share_bounds(2, 4) == [(0, 0), (0, 1), (1, 1), (1, 2)]
    totals = [(100, 3), (50, 0), (70, 2)]
    base0, R0, d_in, draws = shard.place_shares(totals, None)
    assert (base0, R0, d_in, draws) == ([0, 100, 150], [0, 3, 3], None, None)
    # share 0 starts at deviation 0 and leaves 2; share 1 has no draw-dependent gap (identity on 0 .. 3); share 2 maps 2 -> 4
    fns = [[2], [0, 1, 2, 3], [0, 1, 4, 5]]
    base0, R0, d_in, draws = shard.place_shares(totals, fns)
    assert d_in == [0, 2, 2] and draws == 100 + 50 + 70 + 4
    # a deviation beyond what a function lists reads its last entry (the C side clamps the same way)
    assert shard.place_shares([(10, 1), (10, 0)], [[5], [7]])[2:] == ([0, 5], 20 + 7)
    c = shard.LocalComm()
    assert c.all_gather([1, 2, 3]) == [[1, 2, 3]] and (c.rank, c.world) == (0, 1)


@pytest.mark.parametrize("seed,skip", [(1, 0), (1, 1), (42, 30), (7, 31), (1, 4095), (20240101, 5000000), (3, (1 << 33) + 12345)])
def test_a_jump_over_values_drawn_elsewhere_lands_in_the_same_stream(product, seed, skip):
    """g2s_share_end moves a rank's generator past the whole list — drawn on other ranks' devices — by the
    recurrence's jump polynomial; the stream behind it is the one reached by drawing (checked against libc's rand()
    by test_host.py) wherever materialising that many values is feasible, and consistent with two shorter jumps beyond."""
    n = 200
    if skip <= 6000000:
        assert product.test_rand_skip(seed, skip, n) == product.test_rand_stream(seed, skip, n)
    else:
        a = product.test_rand_skip(seed, skip, n)
        assert a[100:] == product.test_rand_skip(seed, skip + 100, 100)
        assert a != product.test_rand_skip(seed, skip + 1, n)


def test_dist_comm_all_gather_two_ranks_gloo(tmp_path):
    """The exchange fill_share runs between the steps of a share: lists of integers of different lengths per rank,
    over gloo, every rank receiving every rank's."""
    script = tmp_path / "ag.py"
    script.write_text(
        "import os, sys, json\n"
        "sys.path.insert(0, %r)\n"
        "import torch.distributed as dist\n"
        "from gap2seq_amd import shard\n"
        "dist.init_process_group('gloo')\n"
        "c = shard.DistComm(dist)\n"
        "mine = [c.rank + 1] * (3 + 4 * c.rank) + [1 << 40]\n"
        "got = c.all_gather(mine)\n"
        "empty = c.all_gather([] if c.rank else [9])\n"
        "fixed = c.all_gather([7, c.rank], 5)\n"
        "print(json.dumps({'rank': c.rank, 'got': got, 'empty': empty, 'fixed': fixed}))\n"
        "dist.destroy_process_group()\n" % ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    rows = [json.loads(ln) for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert sorted(r["rank"] for r in rows) == [0, 1]
    want = [[1, 1, 1, 1 << 40], [2] * 7 + [1 << 40]]
    for r in rows:
        assert r["got"] == want and r["empty"] == [[9], []] and r["fixed"] == [[7, 0], [7, 1]]
