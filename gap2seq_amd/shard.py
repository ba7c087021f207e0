"""gap2seq_amd/shard.py — how a gap list is spread over the GPUs of one node.

Gaps are independent given the read-only graph (fill_gap takes `const Graph&`,
/root/reference/src/Gap2Seq.cpp:858; the reference itself parallelises per gap,
:296-306), so there is NO data-path collective: the graph is replicated in every
GPU's HBM, the gap list is cut into contiguous chunks, chunk c belongs to rank
c mod world (static), and ranks that finish early take chunks from the tail of
the most loaded rank's share (host-side stealing; in one-process-per-GPU mode the
steal order is decided up front from the per-chunk cost estimate, so no
communication is needed).  torch.distributed is only used by bench.py for the
barrier and the max-over-ranks timing.
"""


def chunk_bounds(n_items, chunk):
    """Contiguous chunks [(begin, end), ...] of at most `chunk` items."""
    if chunk <= 0:
        raise ValueError("chunk must be positive")
    return [(b, min(n_items, b + chunk)) for b in range(0, n_items, chunk)]


def gap_cost(gap_len, d_err, lmf, rmf):
    """Levels the DP runs for one gap: D = lmf + rmf + g + e (Gap2Seq.cpp:862-863,1029)."""
    return lmf + rmf + gap_len + d_err


def assign_chunks(costs, world):
    """Static round-robin start, then greedy stealing: repeatedly move the last chunk
    of the most loaded rank to the least loaded rank while that lowers the maximum.
    costs: per-chunk cost estimates.  Returns a list of chunk-index lists per rank;
    every rank keeps its chunks in increasing order so results merge by index."""
    if world <= 0:
        raise ValueError("world must be positive")
    owner = [[] for _ in range(world)]
    for c in range(len(costs)):
        owner[c % world].append(c)
    load = [sum(costs[c] for c in o) for o in owner]
    while True:
        hi = max(range(world), key=lambda r: load[r])
        lo = min(range(world), key=lambda r: load[r])
        if hi == lo or not owner[hi]:
            break
        c = owner[hi][-1]
        if max(load[hi] - costs[c], load[lo] + costs[c]) >= load[hi]:
            break
        owner[hi].pop()
        owner[lo].append(c)
        load[hi] -= costs[c]
        load[lo] += costs[c]
    for o in owner:
        o.sort()
    return owner


def shard_for_rank(n_items, costs_per_item, rank, world, chunk=64):
    """Item indices (sorted) that `rank` processes."""
    bounds = chunk_bounds(n_items, chunk)
    costs = [sum(costs_per_item[b:e]) for b, e in bounds]
    mine = assign_chunks(costs, world)[rank]
    idx = []
    for c in mine:
        idx.extend(range(*bounds[c]))
    return idx


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
