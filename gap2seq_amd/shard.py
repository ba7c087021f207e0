"""gap2seq_amd/shard.py — how bench.py spreads one gap list over the GPUs of a node.

Gaps are independent given the read-only graph (fill_gap takes `const Graph&`,
/root/reference/src/Gap2Seq.cpp:858; the reference itself parallelises per gap,
:296-306), so there is NO data-path collective: the graph is replicated in every
GPU's HBM and the list is cut into contiguous groups that the sessions of
g2s_team_fill (one host thread + stream per GPU, in ONE process) pull from a shared
counter: a static start with stealing by construction.  This module only chooses
the group size and reduces the timing; torch.distributed is used for nothing but
the barriers when bench.py is started by torch.distributed.run.
"""


def group_size(n_gaps, n_sessions, min_group=256, per_session=1):
    """Gaps per group for g2s_team_fill.  The sessions pull groups from one counter — whoever is
    free takes the next one (tests/test_shard.py shows a slow worker's share being taken over) —
    so `per_session` > 1 groups per session balance devices of unequal speed.  The default is ONE
    group per session: a launch of the fill kernel is bound by its slowest gap (a dependent chain
    of search rounds), not by the number of gaps, and a session runs its groups one after the
    other — measured on config 3's list with the sessions sharing one MI355X
    (profiles/r03_shared_device_sessions.txt): four groups per session take 2-3x as long as one.
    Never fewer than `min_group` gaps per group (some sessions may then stay idle on short lists)."""
    if n_gaps <= 0 or n_sessions <= 0 or per_session <= 0:
        raise ValueError("n_gaps, n_sessions and per_session must be positive")
    return max(min(min_group, n_gaps), -(-n_gaps // (n_sessions * per_session)))


def group_bounds(n_gaps, group):
    """The contiguous groups [(begin, end), ...] g2s_team_fill cuts a list into."""
    if group <= 0:
        raise ValueError("group must be positive")
    return [(b, min(n_gaps, b + group)) for b in range(0, n_gaps, group)]


def reduce_timing(seconds, units, dist=None):
    """(max seconds over ranks, total units over ranks).  `dist` is torch.distributed
    or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds, units
    import torch
    t = torch.tensor([seconds], dtype=torch.float64)
    u = torch.tensor([units], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t, u = t.cuda(), u.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
