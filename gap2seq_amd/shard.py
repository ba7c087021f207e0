"""gap2seq_amd/shard.py — how bench.py spreads one gap list over the GPUs of a node.

Gaps are independent given the read-only graph (fill_gap takes `const Graph&`,
/root/reference/src/Gap2Seq.cpp:858; the reference itself parallelises per gap,
:296-306), so there is NO data-path collective: the graph is replicated in every
GPU's HBM and the list is cut into contiguous groups that the sessions of
g2s_team_fill (one host thread + stream per GPU, in ONE process) pull from a shared
counter: a static start with stealing by construction.  This module only chooses
the group size and reduces the timing; torch.distributed is used for nothing but
the barriers when bench.py is started by torch.distributed.run.

Round 6: a launcher that pins ONE device per rank (HIP_VISIBLE_DEVICES per rank) leaves that
one process a single GPU.  `fill_share` is the same partitioning across PROCESSES: rank r
fills, traces and writes the r-th contiguous share of the list on its own GPU (g2s_share_*,
include/g2s.h), and the shares are placed in the reference's one rand() stream
(Gap2Seq.cpp:178,1440,1513: draws in input order) by two all-gathers of host scalars —
every share's draw totals, then every share's function "deviation behind me for deviation
in front of me".  Still no data-path collective: nothing crosses between the devices.
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


def share_bounds(n_gaps, world):
    """The contiguous shares [(begin, end), ...] of a list over `world` ranks (the sizes differ by at most one)."""
    if n_gaps < 0 or world <= 0:
        raise ValueError("n_gaps must not be negative, world must be positive")
    return [(n_gaps * r // world, n_gaps * (r + 1) // world) for r in range(world)]


def place_shares(totals, fns):
    """Where every share begins in the list's one rand() stream.  totals[r] = (draws of share r if every
    draw-dependent gap took its fewest, its summed spreads); fns[r] = share r's function: fns[r][d] = the deviation
    behind share r when the deviation in front of it is d, d = 0 .. R0_r (None while only the bases are wanted).
    Returns (base0, R0, d_in, list_draws): per rank the sums over the ranks in front, the deviation each share starts
    with (the functions of the ranks in front composed from 0), and what the whole list drew."""
    world = len(totals)
    base0, R0 = [0] * world, [0] * world
    for r in range(1, world):
        base0[r] = base0[r - 1] + int(totals[r - 1][0])
        R0[r] = R0[r - 1] + int(totals[r - 1][1])
    if fns is None:
        return base0, R0, None, None
    d_in, d = [0] * world, 0
    for r in range(world):
        d_in[r] = d
        f = fns[r]
        d = int(f[min(d, R0[r], len(f) - 1)])
    return base0, R0, d_in, sum(int(t[0]) for t in totals) + d


class LocalComm:
    """all_gather for ONE rank (world 1), and the interface fill_share expects of a communicator."""
    rank, world = 0, 1

    def all_gather(self, values, maxlen=None):
        return [list(values)]


class DistComm:
    """all_gather of a list of non-negative integers over torch.distributed (any backend; the values are host scalars —
    with the nccl backend they take a detour through the device, which this path has no use for: use gloo)."""

    def __init__(self, dist):
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()

    def all_gather(self, values, maxlen=None):
        """maxlen: no rank's list is longer, and every rank passes the same number — ONE collective (every list travels
        behind its length); without it the lengths are exchanged first."""
        import torch
        dist = self.dist
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        if maxlen is None:
            n = torch.tensor([len(values)], dtype=torch.int64, device=dev)
            ns = [torch.zeros_like(n) for _ in range(self.world)]
            dist.all_gather(ns, n)
            maxlen = max(int(x.item()) for x in ns)
        if len(values) > maxlen:
            raise ValueError("all_gather: %d values, at most %d announced" % (len(values), maxlen))
        mine = torch.zeros(1 + max(1, maxlen), dtype=torch.int64, device=dev)
        mine[0] = len(values)
        if len(values):
            mine[1:1 + len(values)] = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine)
        rows = [o.tolist() for o in out]
        return [[int(v) for v in row[1:1 + int(row[0])]] for row in rows]


def fill_share(P, session, comm, arr, n, results, arena_ptr, arena_bytes):
    """This rank's share of a list (a ctypes array `arr` of n g2s_gap, results / arena in g2s_host_alloc memory) through
    g2s_share_begin / _tables / _trace / _end with the two exchanges in between.  Returns the list's draws, or None
    when some rank cannot take its share this way (every rank then leaves its share untouched and the caller falls back)."""
    import ctypes as C
    lib = P.load_library()
    tot = (C.c_uint64 * 2)()
    rc = lib.g2s_share_begin(session.h, arr, n, results, arena_ptr, arena_bytes, tot)
    if rc not in (0, P.G2S_ERR_STATE):
        P._check(rc)
    all_tot = comm.all_gather([1 if rc == 0 else 0, int(tot[0]), int(tot[1])], 3)
    if not all(t[0] for t in all_tot):
        if rc == 0:
            lib.g2s_share_end(session.h, 0)  # (drops the share: it was not traced)
        return None
    totals = [(t[1], t[2]) for t in all_tot]
    base0, R0, _, _ = place_shares(totals, None)
    fn = C.POINTER(C.c_uint32)()
    rc = lib.g2s_share_tables(session.h, base0[comm.rank], R0[comm.rank], C.byref(fn))
    if rc not in (0, P.G2S_ERR_STATE):
        P._check(rc)
    mine = [1] + [int(fn[d]) for d in range(R0[comm.rank] + 1)] if rc == 0 else [0]
    all_fn = comm.all_gather(mine, max(R0) + 2)  # (every rank knows every share's R0: the longest function)
    if not all(f[0] for f in all_fn):
        if rc == 0:
            lib.g2s_share_end(session.h, 0)
        return None
    _, _, d_in, list_draws = place_shares(totals, [f[1:] for f in all_fn])
    rc = lib.g2s_share_trace(session.h, d_in[comm.rank])
    ok = comm.all_gather([1 if rc == 0 else 0], 1)
    if rc not in (0, P.G2S_ERR_STATE):
        P._check(rc)
    if not all(o[0] for o in ok):
        if rc == 0:
            lib.g2s_share_end(session.h, 0)
        return None
    P._check(lib.g2s_share_end(session.h, list_draws))
    return list_draws
