"""tests/seg_model.py — executable model of the algorithm of the segment tier (kernel
g2s_fill_seg, gap2seq_amd/csrc/fill_seg.hip).  TEST INFRASTRUCTURE: it exists so that the
algorithm itself — not only its HIP implementation — is pinned against the CPU oracle on a
machine without a GPU, and so that the kernel has a line-by-line restatement to be read
against.  It is not a fill path: it is never imported by gap2seq_amd/.

What it models (phases of /root/reference/src/Gap2Seq.cpp:858-1312):

  phase B   The left DP as a search over SEGMENTS instead of levels.  Node ids are numbered
            along unitigs (dbg.hpp): inside a unitig the only successor of an even id v is
            v+2, of an odd id v-2, and v is that node's only predecessor.  So a state
            (v, d, count) entered at a unitig boundary determines the whole diagonal
            (v +- 2t, d + t, count) up to the unitig's end, the last DP level D, or the first
            state the pruning rule (:1050) rejects.  Only ENTRY EVENTS (node, depth) carry
            counts that several parents add up; they are kept in a table keyed (node, depth)
            and expanded in batches: an event is final once no pending event can still create
            a child at its depth or below, i.e. its depth is below the HORIZON
            H = min over pending events of (depth + steps to the end of its unitig + 1).
            Left-flank seeds (:1082-1105, row value ASSIGNED 1: Q6) are pre-inserted events
            with a fixed count; events above the flank (depth < lmf) are cut after one state so
            that no segment runs across a seed state.
  phase C   (:1107-1159) evaluated from the target hits in closed form: a hit (j, depth) is
            found at level |depth - (g+lmf+j)| + g+lmf+rmf; the smallest level wins, then the
            smallest j.
  Q7        both strands of a k-mer at one depth: among events (same table slot), and where
            an upward and a downward segment of one unitig cross (arithmetic on the segments).
  phase D1  (:1169-1312) the backward closure over segments, generation by generation in
            reverse (a segment's children were all created after it was expanded), from the
            sink states (:1195-1244, Q3/Q4) and the traceback starts (:1245-1259); then the
            closure is expanded into the 16-byte per-state records the host part of phase D
            takes, children before parents.
"""
import numpy as np

INVALID = 0xFFFFFFFF
MAX_PATHS = 2147483647 // 2 - 1
SUB_IN_S, SUB_IN_T, SUB_SOURCE, SUB_SINK, SUB_START_T = 1, 2, 4, 8, 16
META_FLAG_SHIFT = 27
SUB_MORE = 0x40000000


class Tables:
    """The graph as the kernels see it: successor table + steps left inside the unitig."""

    def __init__(self, product, graph):
        n = graph.num_kmers
        succ, words = product.test_graph_tables(graph)
        self.n = n
        self.succ = np.frombuffer(succ, dtype=np.uint32, count=8 * n).reshape(2 * n, 4).copy()
        w = np.frombuffer(words, dtype=np.uint64, count=(n + 63) // 64)
        bits = np.unpackbits(w.view(np.uint8), bitorder="little")[:n]
        starts = np.flatnonzero(bits)
        idx = np.arange(n)
        pos = np.searchsorted(starts, idx, side="right")
        start_of = starts[pos - 1]
        nxt = np.append(starts, n)[pos]
        self.rem = np.empty(2 * n, dtype=np.int64)  # internal steps available from an oriented node
        self.rem[0::2] = nxt - 1 - idx               # even orientation walks up
        self.rem[1::2] = idx - start_of              # odd orientation walks down

    def successors(self, v):
        return [int(w) for w in self.succ[v] if w != INVALID]


def seg_node(v0, t):
    return v0 + 2 * t if (v0 & 1) == 0 else v0 - 2 * t


def seg_pos(v0, length, node):
    """t with seg_node(v0, t) == node and 0 <= t < length, else -1."""
    if node == INVALID or (node ^ v0) & 1:
        return -1
    t = (node - v0) // 2 if (v0 & 1) == 0 else (v0 - node) // 2
    return t if 0 <= t < length else -1


class Gap:
    def __init__(self, g, e, lmf, rmf, lseeds, rseeds, targets, all_paths=True, skip_confident=False):
        self.g, self.e, self.lmf, self.rmf = g, e, lmf, rmf
        self.lseeds, self.rseeds, self.targets = lseeds, rseeds, targets
        self.all_paths, self.skip_confident = all_paths, skip_confident
        self.D = lmf + rmf + g + e                    # :862-863
        self.right_half = rmf + (g + e + 1) // 2      # :862
        self.prune_from = g // 2 + e // 2 + lmf       # :1050


def right_set(tb, gap):
    """Phase A (:871-982): k-mer indices within right_half predecessor steps of a right-flank
    seed (seed j enters at depth j).  The tier keeps phase A of the LDS tier; only the set matters."""
    INF = 1 << 60
    depth = {}
    frontier = []
    d = 0
    seeds = gap.rseeds
    if seeds[0] != INVALID:
        depth[seeds[0]] = 0
        frontier = [seeds[0]]
    while d < gap.right_half:
        d += 1
        nxt = []
        for v in frontier:
            for w in tb.successors(v ^ 1):  # predecessors(v)[i] = successors(v^1)[i] ^ 1
                p = w ^ 1
                if depth.get(p, INF) > d:
                    if p not in depth:
                        nxt.append(p)
                    depth[p] = d
        frontier = nxt
        if d <= gap.rmf and seeds[d] != INVALID and seeds[d] not in depth:
            depth[seeds[d]] = d
            frontier.append(seeds[d])
    return {v >> 1 for v in depth}


class Result:
    pass


def fill_model(tb, gap, rs=None, stats=None):
    """Phases B, C, Q7, D1 of one gap.  Returns Result with: states [(node, depth, count)],
    xB, sB, c_count, lengths, reached_j, final_d, q7, records [(node, cnt, meta, pred)], xp."""
    if rs is None:
        rs = right_set(tb, gap)
    D, lmf, rmf, g, e = gap.D, gap.lmf, gap.rmf, gap.g, gap.e
    # ---- phase B ------------------------------------------------------------------------------
    events = {}   # (node, depth) -> [count, fixed, [parent segments]]
    pending = set()
    q7 = False
    by_kmer = {}  # (k-mer, depth) -> node of the first event there

    def add_event(node, depth, count, parent, fixed=False):
        nonlocal q7
        key = (node, depth)
        ev = events.get(key)
        if ev is None:
            ev = events[key] = [0, False, []]
            pending.add(key)
            other = by_kmer.setdefault((node >> 1, depth), node)
            if other != node:
                q7 = True
        if fixed:
            ev[1] = True
        else:
            ev[0] = min(MAX_PATHS, ev[0] + count)  # saturating add is associative: the order does not matter
            ev[2].append(parent)

    segs = []     # [v0, d0, count, length, parents, generation]
    gen = 0
    s0 = gap.lseeds[0]
    chain = (lmf >= 1 and s0 != INVALID and all(gap.lseeds[d] == seg_node(s0, d) for d in range(lmf + 1))
             and int(tb.rem[s0]) >= lmf
             and all(seg_pos(s0, lmf, t) < 0 for t in gap.targets))  # (no sink / traceback start inside the chain)
    if chain:
        # the usual flank: seed d is the d-th node after seed 0 inside one unitig.  Levels 0 .. lmf-1 are
        # that chain with count 1: one segment, and the seed at depth lmf as the only pending event
        segs.append([s0, 0, 1, lmf, [], 0])
        gen = 1
        add_event(gap.lseeds[lmf], lmf, 0, None, fixed=True)
        events[(gap.lseeds[lmf], lmf)][2].append(0)
    else:
        for d in range(lmf + 1):
            if gap.lseeds[d] != INVALID and d <= D:
                add_event(gap.lseeds[d], d, 0, None, fixed=True)
    rounds = 0
    max_pending = max_batch = 0
    while pending:
        max_pending = max(max_pending, len(pending))
        def steps(key):
            node, depth = key
            return 1 if depth < lmf else int(tb.rem[node]) + 1  # states up to the end of the unitig
        horizon = min(depth + steps((node, depth)) for node, depth in pending)
        batch = sorted(k for k in pending if k[1] < horizon)
        assert batch
        rounds += 1
        max_batch = max(max_batch, len(batch))
        for key in batch:
            pending.discard(key)
            node, depth = key
            cnt, fixed, parents = events[key]
            if fixed:
                cnt, parents = 1, [p for p in parents]
            lcap = min(steps(key), D - depth + 1)
            length = 1
            while length < lcap:  # interior states are entered under the pruning rule (:1050)
                dd = depth + length
                if dd >= gap.prune_from and (seg_node(node, length) >> 1) not in rs:
                    break
                length += 1
            sid = len(segs)
            segs.append([node, depth, cnt, length, parents, gen])
            x_depth = depth + length - 1
            if length == lcap and x_depth < D:  # the walk reached the end of its stretch: leave through the table
                for w in tb.successors(seg_node(node, length - 1)):
                    if x_depth + 1 < gap.prune_from or (w >> 1) in rs:
                        add_event(w, x_depth + 1, cnt, sid)
        gen += 1
    n_gen = gen
    # ---- Q7 among segments: an upward and a downward segment of one unitig meeting on a k-mer ----
    ups = [s for s in segs if (s[0] & 1) == 0 and s[3] > 0]
    downs = [s for s in segs if (s[0] & 1) == 1 and s[3] > 0]
    if not q7 and ups and downs:
        for a in ups:
            ia, da, la = a[0] >> 1, a[1], a[3]
            for b in downs:
                ib, db, lb = b[0] >> 1, b[1], b[3]
                s_, dl = ib - ia, db - da
                if (s_ + dl) & 1:
                    continue
                t1, t2 = (s_ + dl) // 2, (s_ - dl) // 2
                if 0 <= t1 < la and 0 <= t2 < lb:
                    q7 = True
                    break
            if q7:
                break
    # ---- phase C in closed form -----------------------------------------------------------------
    best = None  # (found level, j) -> [c1, c2]
    c12 = [0, 0]
    for v0, d0, cnt, length, _, _ in segs:
        for j in range(rmf + 1):
            t = seg_pos(v0, length, gap.targets[j])
            if t < 0:
                continue
            td = d0 + t
            base = g + lmf + j
            err = abs(td - base)
            if err > e:
                continue
            key = (err + g + lmf + rmf, j)
            if best is None or key < best:
                best, c12 = key, [0, 0]
            if key == best:
                c12[0 if td >= base else 1] = cnt
    res = Result()
    res.q7 = q7
    res.rounds, res.n_seg, res.n_gen = rounds, len(segs), n_gen
    res.max_pending, res.max_batch = max_pending, max_batch
    found = best is not None
    res.c_count = min(MAX_PATHS, c12[0] + c12[1]) if found else 0
    res.reached_j = best[1] if found else 0
    res.lengths = []
    if found:
        err = best[0] - (g + lmf + rmf)
        l1, l2 = g + lmf + best[1] + err, g + lmf + best[1] - err
        if c12[0] > 0:
            res.lengths.append(l1)
            if c12[1] > 0:
                res.lengths.append(l2)
        else:
            res.lengths.append(l2)
    res.final_d = best[0] if (found and not gap.all_paths) else D + 1
    d_last = D if gap.all_paths or not found else best[0]  # -best-only: the DP stops after the level of the find
    res.states = []
    res.xB = res.sB = 0
    for v0, d0, cnt, length, _, _ in segs:
        res.sB += max(0, min(length, d_last - d0 + 1))
        res.xB += max(0, min(length, d_last - d0))
    res.segs = segs
    res.d_last = d_last
    res.records, res.xp = [], []
    if not (res.c_count > 0 and res.lengths):
        return res
    # ---- phase D1 over segments -------------------------------------------------------------------
    want_s = not gap.skip_confident
    sinknode = gap.targets[rmf - 1] if (want_s and gap.all_paths and rmf >= 1) else INVALID
    lo_sink = max(0, lmf + g - e)  # :1196
    reached = gap.targets[res.reached_j]
    t_flags = SUB_IN_T | SUB_START_T | ((SUB_IN_S | SUB_SINK) if (want_s and not gap.all_paths) else 0)
    ns = len(segs)
    t_s = [-1] * ns   # last state of the segment on a path to a sink (-1: none)
    t_t = [-1] * ns   # last state reachable backwards from a traceback start
    child_s = [False] * ns
    child_t = [False] * ns

    def is_source(node, depth):
        return depth <= lmf and gap.lseeds[depth] != INVALID and (node >> 1) == (gap.lseeds[depth] >> 1)

    for sid in range(ns - 1, -1, -1):  # (the kernel sweeps generation by generation; any order with children first)
        v0, d0, cnt, length, parents, _ = segs[sid]
        length = max(0, min(length, d_last - d0 + 1))
        if length == 0:
            continue
        own_s = own_t = -1
        ts = seg_pos(v0, length, sinknode)
        if ts >= 0 and d0 + ts >= lo_sink:
            own_s = ts
        tt = seg_pos(v0, length, reached)
        if tt >= 0 and (d0 + tt) in res.lengths:
            own_t = tt
            if t_flags & SUB_IN_S:
                own_s = max(own_s, tt)
        t_s[sid] = length - 1 if child_s[sid] else own_s
        t_t[sid] = length - 1 if child_t[sid] else own_t
        if t_s[sid] < 0 and t_t[sid] < 0:
            continue
        if d0 > 0 and not is_source(v0, d0):
            for p in parents:
                child_s[p] = child_s[p] or t_s[sid] >= 0
                child_t[p] = child_t[p] or t_t[sid] >= 0
    # ---- emission: children before parents, inside a segment from its last closure state down ----
    base = [0] * ns
    n_rec = 0
    for sid in range(ns - 1, -1, -1):
        base[sid] = n_rec
        n_rec += max(t_s[sid], t_t[sid]) + 1
    for sid in range(ns - 1, -1, -1):
        v0, d0, cnt, length, parents, _ = segs[sid]
        tmax = max(t_s[sid], t_t[sid])
        for t in range(tmax, -1, -1):
            node, depth = seg_node(v0, t), d0 + t
            f = (SUB_IN_S if t <= t_s[sid] else 0) | (SUB_IN_T if t <= t_t[sid] else 0)
            if node == sinknode and depth >= lo_sink:
                f |= SUB_IN_S | SUB_SINK
            if node == reached and depth in res.lengths:
                f |= t_flags
            pred = -1
            if t > 0:
                pred = base[sid] + (tmax - t) + 1
            else:
                if is_source(node, depth):
                    f |= SUB_SOURCE
                elif depth > 0 and parents:
                    ps = sorted(set(parents))
                    pred = base[ps[0]]  # the parent's last state: a child in the closure puts all of it there
                    for p in ps[1:]:
                        pred |= SUB_MORE
                        res.xp.append(((base[sid] + tmax) << 32) | base[p])
            res.records.append((node, min(cnt, MAX_PATHS), depth | (f << META_FLAG_SHIFT), pred))
    assert len(res.records) == n_rec
    # the same closure as the kernel emits it: one 8-word record per segment with closure states,
    # children before parents (descending segment id), parents as indices among the emitted segments
    emitted = [sid for sid in range(ns - 1, -1, -1) if max(t_s[sid], t_t[sid]) >= 0]
    index = {sid: i for i, sid in enumerate(emitted)}
    res.compact = []
    for sid in emitted:
        v0, d0, cnt, length, parents, _ = segs[sid]
        src = is_source(v0, d0)
        ps = [0xFFFF] * 4
        if not src and d0 > 0:
            for q, pp in enumerate(sorted(set(parents))):
                ps[q] = index[pp]
        res.compact.append((v0, d0 | ((max(t_s[sid], t_t[sid]) + 1) << 16), min(cnt, MAX_PATHS),
                            (t_s[sid] if t_s[sid] >= 0 else 0x7FFF) | ((t_t[sid] if t_t[sid] >= 0 else 0x7FFF) << 16),
                            ps[0] | (ps[1] << 16), ps[2] | (ps[3] << 16),
                            SUB_SOURCE if src else 0, 0))
    if stats is not None:
        stats.append((rounds, len(segs), n_gen))
    return res


def expand_states(res):
    """[(node, depth, count)] of the DP table the segments stand for (what the oracle dumps)."""
    out = []
    for v0, d0, cnt, length, _, _ in res.segs:
        for t in range(max(0, min(length, res.d_last - d0 + 1))):
            out.append((seg_node(v0, t), d0 + t, cnt))
    return out


def right_entries(tb, gap):
    """Phase A as the tier runs it: a label-correcting search over unitigs.  An entry
    (node, label) covers its unitig backwards (predecessor direction) for
    min(steps left in the unitig, right_half - label) steps; where the unitig ends with budget
    left, the predecessors of its last node are proposed with label + steps + 1.  The right set
    is the union of the covered index intervals; only the k-mer (not the strand) matters (:1050).
    Returns {node: label}."""
    label = {}
    queue = []
    for j, s in enumerate(gap.rseeds):
        if s != INVALID and j <= gap.right_half and label.get(s, 1 << 60) > j:
            label[s] = j
            queue.append(s)
    while queue:
        nxt = []
        for v in queue:
            d = label[v]
            # predecessors of v = successors of v^1, flipped: walking back from v is walking on from v^1
            steps = min(int(tb.rem[v ^ 1]), gap.right_half - d)
            last = seg_node(v ^ 1, steps) ^ 1
            if d + steps < gap.right_half:
                for w in tb.successors(last ^ 1):
                    p = w ^ 1
                    if label.get(p, 1 << 60) > d + steps + 1:
                        label[p] = d + steps + 1
                        nxt.append(p)
        queue = list(dict.fromkeys(nxt))
    return label


def entries_to_set(tb, gap, label):
    out = set()
    for v, d in label.items():
        steps = min(int(tb.rem[v ^ 1]), gap.right_half - d)
        for t in range(steps + 1):
            out.add(seg_node(v ^ 1, t) >> 1)
    return out
