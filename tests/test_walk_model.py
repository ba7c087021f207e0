"""The two formulations of a traceback over closure segments (gap2seq_amd/csrc/d3_device.hip), as executable models.

The reference's traceback (Gap2Seq.cpp:1437-1522) goes state by state and draws one rand() value per state it passes
(:1513 draws for a single parent too).  Over SEGMENTS — (entry depth d0, length L, parents, source flag); state t of a
segment sits at depth d0 + t; a child in the closure puts the whole parent there — the device walked with counters
(`d3_walk_count`, and g2s_d3_trace's walk until round 4): states left in the segment, depth, draws.  Round 4's kernels
use what follows from "one draw per state passed": the draw made at a segment's first state is the (1 + len - d0)-th of
the gap whatever the path, so (i) the parent taken from a segment is a property of the segment, (ii) the walk proper is
the CHAIN of those parents, (iii) depths, checks and the draw count (1 + len - the depth the walk ends at) follow from
the chain by a scan.  These tests pin that the two agree on random closures — results on consistent ones, the verdict
"not a walk" on damaged ones — with nothing but Python: the GPU suite pins the kernels against the oracle."""
import random

NOPAR = 0xFFFF


def _closure(rng, len_total, branch=0.5, damage=None):
    """A consistent closure for a traceback of length len_total: segments tile every path from the start (depth
    len_total) down to a source.  Returns (segments, start_seg, start_t); a segment is a dict d0, L, parents, source."""
    segs = []

    def build(top):  # a segment whose LAST state sits at depth `top`; returns its id
        L = rng.randint(1, min(9, top + 1))
        d0 = top - (L - 1)
        me = len(segs)
        segs.append(dict(d0=d0, L=L, parents=[], source=False))
        if d0 == 0 or (rng.random() < 0.15 and d0 <= len_total // 2):
            segs[me]["source"] = True
        else:
            n = 1 if rng.random() > branch or len(segs) > 60 else rng.randint(2, 4)
            segs[me]["parents"] = [build(d0 - 1) for _ in range(n)]
        return me

    L0 = rng.randint(1, min(9, len_total + 1))
    t0 = rng.randint(0, L0 - 1)  # the start is state t0 of segment 0, at depth len_total
    d0 = len_total - t0
    segs.append(dict(d0=d0, L=L0, parents=[], source=False))
    if d0 == 0:
        segs[0]["source"] = True
    else:
        n = 1 if rng.random() > branch else rng.randint(2, 4)
        segs[0]["parents"] = [build(d0 - 1) for _ in range(n)]
    if damage == "depth":  # a segment that does not sit where its child says
        s = segs[rng.randrange(1, len(segs))] if len(segs) > 1 else segs[0]
        s["d0"] += rng.choice([-2, -1, 1, 2])
        s["d0"] = max(0, s["d0"])
    elif damage == "no_way":  # a segment above depth 0 without a parent and without being a source
        cands = [s for s in segs if s["source"] and s["d0"] > 0]
        if cands:
            rng.choice(cands)["source"] = False
    return segs, 0, t0


def _walk_with_counters(segs, start, t0, len_total, values):
    """d3_walk_count / the old walk of g2s_d3_trace: returns (draws, end depth, hops) or None."""
    draws, d2, i, t = 1, len_total, start, t0
    hops = []
    for guard in range(10 * len(segs) + 10):
        if d2 < 0:
            return None
        s = segs[i]
        if s["d0"] + t != d2:
            return None
        hops.append((d2, i, t))
        draws += t
        d2 -= t
        if s["source"]:
            return draws, d2, hops
        if d2 > 0:
            if not s["parents"]:
                return None
            nb = len(s["parents"])
            rv = values[draws] >> 1 if nb > 1 else 0
            draws += 1
            i = s["parents"][rv % nb]
            t = segs[i]["L"] - 1
        d2 -= 1
    return None


def _walk_as_chain(segs, start, t0, len_total, values):
    """Round 4: per segment the parent its first state goes on to (draw index 1 + len - d0), the chain, then a scan."""
    nxt = []
    for s in segs:
        at = 1 + len_total - s["d0"]
        if not s["parents"] or at < 1 or at >= len(values):
            nxt.append(None)
        else:
            nb = len(s["parents"])
            nxt.append(s["parents"][(values[at] >> 1) % nb if nb > 1 else 0])
    chain, i = [], start
    while True:
        if len(chain) >= len(segs):
            return None  # (a traceback descends: it enters a segment once)
        chain.append(i)
        if segs[i]["source"] or nxt[i] is None:
            break
        i = nxt[i]
    if not segs[chain[-1]]["source"]:
        return None
    passed, hops = 0, []
    for h, i in enumerate(chain):
        s = segs[i]
        t = t0 if h == 0 else s["L"] - 1
        at = len_total - passed  # the depth at which this hop is entered
        if s["d0"] + t != at or t < 0 or (h + 1 < len(chain) and s["d0"] < 1):
            return None
        hops.append((at, i, t))
        passed += t + 1
    d_end = segs[chain[-1]]["d0"]
    return 1 + len_total - d_end, d_end, hops


def test_chain_and_counters_agree_on_consistent_closures():
    rng = random.Random(20240104)
    checked = multi = 0
    for case in range(600):
        len_total = rng.randint(0, 60)
        segs, start, t0 = _closure(rng, len_total, branch=rng.choice([0.0, 0.3, 0.7]))
        values = [rng.getrandbits(31) for _ in range(len_total + 3)]
        a = _walk_with_counters(segs, start, t0, len_total, values)
        b = _walk_as_chain(segs, start, t0, len_total, values)
        assert a is not None and a == b, (case, a, b)
        draws, d_end, hops = a
        assert draws == 1 + len_total - d_end and draws <= len(values)
        # every state passed draws once: the draw at a segment's first state is the (1 + len - d0)-th
        assert all(at - t == segs[i]["d0"] for at, i, t in hops)
        checked += 1
        multi += any(len(segs[i]["parents"]) > 1 for _, i, _ in hops)
    assert checked == 600 and multi > 100


def test_chain_and_counters_agree_on_what_is_not_a_walk():
    rng = random.Random(7)
    rejected = 0
    for case in range(600):
        len_total = rng.randint(1, 60)
        segs, start, t0 = _closure(rng, len_total, branch=rng.choice([0.0, 0.5]), damage=rng.choice(["depth", "no_way"]))
        values = [rng.getrandbits(31) for _ in range(len_total + 3)]
        a = _walk_with_counters(segs, start, t0, len_total, values)
        b = _walk_as_chain(segs, start, t0, len_total, values)
        # (the damage may lie off the path the draws take: then both walk it, with the same result)
        assert (a is None) == (b is None), (case, a, b)
        if a is not None:
            assert a == b
        rejected += a is None
    assert rejected > 100
