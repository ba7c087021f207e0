"""oracle/pyref.py — independent, string-based Python restatement of Gap2Seq-core's
fill path, written from SURVEY.md Appendix A/B (NOT from the C++ oracle).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED BY THE REFERENCE: the reference has no
tests or golden vectors and cannot be built here; this file exists so that the C++
oracle (oracle/g2s_oracle.cpp) is checked by a second implementation that shares no
code or data representation with it (k-mers are Python strings, DP rows are dicts).
Only for small inputs (pure-Python loops).

Reference anchors (/root/reference/src/Gap2Seq.cpp): fill_gap :858-1556,
execute :161-438, print_statistics :100-156, MAX_PATHS :38.
"""
import time

MAX_PATHS = 2147483647 // 2 - 1
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}
_CODE2CHAR = "ACTG"  # GATB code (c>>1)&3


def norm(s):
    """GATB codec: any char maps to ACTG by (c>>1)&3."""
    return "".join(_CODE2CHAR[(ord(c) >> 1) & 3] for c in s)


def revcomp(s):
    return "".join(_COMP[c] for c in reversed(s))


_ORD = {"A": 0, "C": 1, "T": 2, "G": 3}


def _key(s):
    return [_ORD[c] for c in s]


def canon(s):
    """(canonical string, strand): strand 0 iff s sorts strictly before its revcomp
    in GATB's A<C<T<G order."""
    r = revcomp(s)
    if _key(s) < _key(r):
        return s, 0
    return r, 1


class GlibcRand:
    """glibc TYPE_3 rand()."""

    def __init__(self, seed=1):
        self.seed(seed)

    def seed(self, s):
        s &= 0xFFFFFFFF
        if s == 0:
            s = 1
        r = [0] * 34
        r[0] = s if s < 2 ** 31 else s - 2 ** 32
        for i in range(1, 31):
            word = r[i - 1]
            hi, lo = int(word / 127773), 0
            # C division truncates toward zero
            hi = abs(word) // 127773 * (1 if word >= 0 else -1)
            lo = word - hi * 127773
            w = 16807 * lo - 2836 * hi
            if w < 0:
                w += 2147483647
            r[i] = w
        self.r = [x & 0xFFFFFFFF for x in r[:31]]
        self.f, self.b = 3, 0
        for _ in range(310):
            self.next()

    def next(self):
        v = (self.r[self.f] + self.r[self.b]) & 0xFFFFFFFF
        self.r[self.f] = v
        self.f = (self.f + 1) % 31
        self.b = (self.b + 1) % 31
        return (v >> 1) & 0x7FFFFFFF


class Graph:
    """Exact solid canonical k-mer set with GATB neighbour order."""

    def __init__(self, seqs, k, solid):
        self.k = k
        cnt = {}
        for s in seqs:
            run = 0
            for i, c in enumerate(s):
                if (ord(c) >> 3) & 1:
                    run = 0
                    continue
                run += 1
                if run >= k:
                    km = canon(norm(s[i - k + 1:i + 1]))[0]
                    cnt[km] = cnt.get(km, 0) + 1
        self.kmers = {km for km, c in cnt.items() if c >= solid}

    def node(self, s):  # buildNode: oriented sequence string
        return norm(s[:self.k])

    def contains(self, x):
        return canon(x)[0] in self.kmers

    def succ(self, x):
        return [y for y in (x[1:] + nt for nt in "ACTG") if self.contains(y)]

    def pred(self, x):
        # revcomp, append A,C,T,G  ==  prepend T,G,A,C
        return [y for y in (nt + x[:-1] for nt in "TGAC") if self.contains(y)]


def sat(a, b):
    return min(MAX_PATHS, a + b)


class Info:
    def __init__(self):
        self.q7 = 0
        self.draws = 0
        self.phaseC_count = 0
        self.lengths = []
        self.sub = None
        self.backtrace_failed = 0
        self.ctr = dict(xA=0, sA=0, xB=0, sB=0, xD=0, sD=0)
        self.log = ""


def _border_add(border, x, info):
    """border: dict canonical -> oriented string (first insertion wins, Q7)."""
    c = canon(x)[0]
    if c in border:
        if border[c] != x:
            info.q7 = 1
        return False
    border[c] = x
    return True


def fill_gap(G, rng, kl, kr, g, e, lmf, rmf, skip_confident, all_paths, want_fill=True, info=None):
    """Returns (count, left_fuz, right_fuz, fill_bytes_or_None, substats_or_None).
    fill is a list of chars of length L (None entries = unwritten)."""
    k = G.k
    info = info or Info()
    if lmf < 0 or rmf < 0 or len(kl) < k + lmf or len(kr) < k + rmf:
        return 0, 0, 0, None, None
    right_half = rmf + (g + e + 1) // 2
    left_half = lmf + (g + e) // 2
    D = left_half + right_half

    # phase A: right set (only membership by canonical k-mer is consumed)
    right_rows = {}  # oriented string -> set(depths)
    right_kmers = set()
    border = {}

    def mark_right(x, d):
        s = right_rows.setdefault(x, set())
        if d not in s:
            s.add(d)
            info.ctr["sA"] += 1
        right_kmers.add(canon(x)[0])

    x = G.node(kr[len(kr) - k:])
    d = 0
    if G.contains(x):
        _border_add(border, x, info)
        mark_right(x, 0)
    while d < right_half:
        d += 1
        nxt = {}
        for n in border.values():
            info.ctr["xA"] += 1
            for p in G.pred(n):
                mark_right(p, d)
                _border_add(nxt, p, info)
        border = nxt
        if d <= rmf:
            x = G.node(kr[len(kr) - k - d:])
            if G.contains(x):
                _border_add(border, x, info)
                mark_right(x, d)

    # phase B + C
    rows = {}  # oriented string -> {depth: count}

    def setrow(x, d, v):
        r = rows.setdefault(x, {})
        if d not in r:
            info.ctr["sB"] += 1
        r[d] = v

    def getrow(x, d):
        return rows.get(x, {}).get(d, 0)

    def in_left(x):  # keyed by canonical k-mer
        return x in rows or revcomp(x) in rows

    border = {}
    x = G.node(kl)
    if G.contains(x):
        _border_add(border, x, info)
        setrow(x, 0, 1)
    count = 0
    lengths = []
    target = None
    reached_fuz = 0
    prune_from = g // 2 + e // 2 + lmf
    d = 1
    while d <= D:
        nxt = {}
        for n in border.values():
            info.ctr["xB"] += 1
            np_ = getrow(n, d - 1)
            for v in G.succ(n):
                if d < prune_from or canon(v)[0] in right_kmers:
                    setrow(v, d, sat(getrow(v, d), np_))
                    _border_add(nxt, v, info)
        border = nxt
        if d <= lmf:
            x = G.node(kl[d:])
            if G.contains(x):
                _border_add(border, x, info)
                setrow(x, d, 1)
        if not lengths and d >= g + lmf + rmf:
            err = d - g - lmf - rmf
            for j in range(rmf + 1):
                if count != 0:
                    break
                target = G.node(kr[j:])
                if not in_left(target):
                    continue
                reached_fuz = j
                l1 = g + lmf + j + err
                l2 = g + lmf + j - err
                v1 = getrow(target, l1)
                if v1 >= 1:
                    count = sat(count, v1)
                    lengths.append(l1)
                if l2 != l1 and l2 >= 0:
                    v2 = getrow(target, l2)
                    if v2 >= 1:
                        count = sat(count, v2)
                        lengths.append(l2)
            if not all_paths and lengths:
                break
        d += 1
    final_d = d
    info.phaseC_count = count
    info.lengths = list(lengths)
    left_fuz = right_fuz = 0
    fill = None
    sub = None

    if count > 0 and lengths and want_fill:
        right_fuz = reached_fuz
        vid = {}
        branch = None
        if not skip_confident:
            # D1: vertices 0 sink, 1 source, then canonical k-mers on demand
            out_e = [[], []]
            in_e = [[], []]

            def vert(x):
                c = canon(x)[0]
                if c not in vid:
                    vid[c] = len(out_e)
                    out_e.append([])
                    in_e.append([])
                return vid[c]

            def add_edge_once(u, v):
                if v not in out_e[u]:
                    out_e[u].append(v)
                    in_e[v].append(u)

            if all_paths:
                count = 0
            back = {}
            d2 = lmf + g + e + rmf
            while d2 >= 0:
                if all_paths:
                    if d2 >= lmf + g - e:
                        for j in range(rmf):
                            rn = G.node(kr[j:])
                            if j < rmf - 1 and G.contains(rn):
                                continue
                            if not in_left(rn):
                                continue
                            v = getrow(rn, d2)
                            if v >= 1:
                                count = sat(count, v)
                                if _border_add(back, rn, info):
                                    info.ctr["sD"] += 1
                                add_edge_once(vert(rn), 0)
                else:
                    for L in lengths:
                        if L == d2:
                            if _border_add(back, target, info):
                                info.ctr["sD"] += 1
                            add_edge_once(vert(target), 0)
                lnode = G.node(kl[d2:]) if d2 <= lmf else None
                nxt = {}
                for cur in back.values():
                    info.ctr["xD"] += 1
                    if d2 > lmf or canon(cur)[0] != canon(lnode)[0]:
                        for p in G.pred(cur):
                            if in_left(p) and getrow(p, d2 - 1) > 0:
                                if _border_add(nxt, p, info):
                                    info.ctr["sD"] += 1
                                add_edge_once(vert(p), vert(cur))
                    else:
                        add_edge_once(1, vert(cur))
                back = nxt
                d2 -= 1

            # D2: SCC (simple recursive-free Kosaraju), contraction with multi-edges
            n_real = len(out_e)
            n_real_edges = sum(len(o) for o in out_e)
            comp = _scc(out_e, in_e)
            ncomp = max(comp) + 1 if comp else 0
            csize = [0] * ncomp
            for c in comp:
                csize[c] += 1
            nontriv = [c for c in range(ncomp) if csize[c] > 1]
            size_nontriv = sum(csize[c] for c in nontriv)
            cnode = {}
            out_m = [list(o) for o in out_e]
            in_m = [list(i) for i in in_e]
            if nontriv:
                for c in range(ncomp):
                    if csize[c] > 1:
                        cnode[c] = len(out_m)
                        out_m.append([])
                        in_m.append([])
                for i in range(n_real):
                    if csize[comp[i]] <= 1:
                        continue
                    for s in in_e[i]:
                        if comp[s] == comp[i]:
                            continue
                        if csize[comp[s]] > 1:
                            if s < i:
                                out_m[cnode[comp[s]]].append(cnode[comp[i]])
                                in_m[cnode[comp[i]]].append(cnode[comp[s]])
                        else:
                            out_m[s].append(cnode[comp[i]])
                            in_m[cnode[comp[i]]].append(s)
                    for t in out_e[i]:
                        if comp[t] == comp[i]:
                            continue
                        if csize[comp[t]] > 1:
                            if t < i:
                                out_m[cnode[comp[i]]].append(cnode[comp[t]])
                                in_m[cnode[comp[t]]].append(cnode[comp[i]])
                        else:
                            out_m[cnode[comp[i]]].append(t)
                            in_m[t].append(cnode[comp[i]])
                members = {i for i in range(n_real) if csize[comp[i]] > 1}
                for v in range(len(out_m)):
                    if v in members:
                        out_m[v] = []
                        in_m[v] = []
                    else:
                        out_m[v] = [t for t in out_m[v] if t not in members]
                        in_m[v] = [s for s in in_m[v] if s not in members]
            loops = 0
            for i in range(n_real):
                if csize[comp[i]] <= 1:
                    c = out_m[i].count(i)
                    if c:
                        loops += c
                        out_m[i] = [t for t in out_m[i] if t != i]
                        in_m[i] = [s for s in in_m[i] if s != i]
            sub = dict(vertices=n_real, edges=n_real_edges - loops, nontrivial=len(nontriv),
                       size_nontrivial=size_nontriv, vertices_final=len(out_m) - size_nontriv,
                       edges_final=sum(len(o) for o in out_m))
            # branch rule over a DFS-based topological order (different from the C++ oracle's Kahn)
            order = _topo(out_m)
            branch = [0] * len(out_m)
            bc = 1
            for v in order:
                din, dout = len(in_m[v]), len(out_m[v])
                if din >= 1 or dout >= 1:
                    if din > 1:
                        bc -= din - 1
                    branch[v] = bc
                    if dout > 1:
                        bc += dout - 1

        # D3: traceback
        L = lengths[rng.next() % len(lengths)]
        info.draws += 1
        d2 = L
        last_solid = d2
        cur = target
        fill = [None] * L
        while d2 >= 0:
            if d2 <= lmf:
                ln = G.node(kl[d2:])
                if canon(ln)[0] == canon(cur)[0]:
                    left_fuz = lmf - d2
                    break
            if d2 > 0:
                if skip_confident:
                    solid = True
                else:
                    solid = branch[vid.get(canon(cur)[0], 0)] == 1
                if solid:
                    last_solid = d2
                ch = cur[-1]
                fill[d2 - 1] = ch.upper() if d2 > last_solid - k else ch.lower()
                cand = [p for p in G.pred(cur) if in_left(p) and getrow(p, d2 - 1) > 0]
                if not cand:
                    info.log += "Unable to backtrace! %d %d %s\n" % (d2, final_d, target)
                    info.backtrace_failed = 1
                    info.sub = sub
                    return 0, left_fuz, right_fuz, fill, sub
                cur = cand[rng.next() % len(cand)]
                info.draws += 1
            d2 -= 1
    info.sub = sub
    return count, left_fuz, right_fuz, fill, sub


def _scc(out_e, in_e):
    """Kosaraju, iterative.  comp ids arbitrary."""
    n = len(out_e)
    seen = [False] * n
    order = []
    for r in range(n):
        if seen[r]:
            continue
        st = [(r, 0)]
        seen[r] = True
        while st:
            v, i = st.pop()
            if i < len(out_e[v]):
                st.append((v, i + 1))
                w = out_e[v][i]
                if not seen[w]:
                    seen[w] = True
                    st.append((w, 0))
            else:
                order.append(v)
    comp = [-1] * n
    c = 0
    for r in reversed(order):
        if comp[r] != -1:
            continue
        st = [r]
        comp[r] = c
        while st:
            v = st.pop()
            for w in in_e[v]:
                if comp[w] == -1:
                    comp[w] = c
                    st.append(w)
        c += 1
    return comp


def _topo(out_m):
    n = len(out_m)
    seen = [False] * n
    post = []
    for r in range(n):
        if seen[r]:
            continue
        st = [(r, 0)]
        seen[r] = True
        while st:
            v, i = st.pop()
            if i < len(out_m[v]):
                st.append((v, i + 1))
                w = out_m[v][i]
                if not seen[w]:
                    seen[w] = True
                    st.append((w, 0))
            else:
                post.append(v)
    return list(reversed(post))


def fill_string(fill, start):
    return "".join(c if c is not None else "?" for c in fill[start:])


def stats_line(comment, filled_start, gap_start, gap_end, paths, buf_tail, k, lmf, rmf, lf, rf,
               skip_confident, unique, sub, gap):
    """print_statistics; buf_tail = &buf[lmf-left_fuz] as str."""
    if paths > 0 and (not unique or paths == 1):
        flen = len(buf_tail) - k
        if not skip_confident:
            up = sum(1 for c in buf_tail[:flen] if c.isupper())
            lo = flen - up
            s = ("Scaffold: %s GapStart: %d GapEnd: %d GapLength: %d PathsFound: %d FilledStart: %d "
                 "FilledEnd: %d FilledGapLength: %d LeftMaxFuz: %d LeftFuz: %d RightMaxFuz: %d RightFuz: %d "
                 "ConfidentBases: %d TotalBases: %d\n" % (comment, gap_start, gap_end, gap, paths, filled_start,
                                                          filled_start + flen, flen, lmf, lf, rmf, rf, up, up + lo))
            s += ("SubgraphStats: Vertices: %d Edges: %d NontrivialStrongComponents: %d "
                  "SizeNontrivialStrongComponents: %d VerticesFinal: %d EdgesFinal: %d\n" %
                  (sub["vertices"], sub["edges"], sub["nontrivial"], sub["size_nontrivial"],
                   sub["vertices_final"], sub["edges_final"]))
            return s
        return ("Scaffold: %s GapStart: %d GapEnd: %d GapLength: %d PathsFound: %d FilledStart: %d FilledEnd: %d "
                "FilledGapLength: %d LeftFuz: %d RightFuz: %d\n" %
                (comment, gap_start, gap_end, gap, paths, filled_start, filled_start + flen, flen, lf, rf))
    s = ("Scaffold: %s GapStart: %d GapEnd: %d GapLength: %d PathsFound: 0 FilledStart: 0 FilledEnd: 0 "
         "FilledGapLength: 0 LeftMaxFuz: %d LeftFuz: %d RightMaxFuz: %d RightFuz: %d" %
         (comment, gap_start, gap_end, gap, lmf, lf, rmf, rf))
    if paths == -1:
        s += " Memory limit exceeded"
    return s + "\n"


def execute_scaffolds(G, records, k, d_err, max_fuz, randseed, skip_confident=False, unique=False,
                      all_paths=True):
    """Scaffold mode (execute :285-438) -> (fasta_text, per-gap log text, filled, gaps)."""
    rng = GlibcRand(randseed if randseed > 0 else int(time.time()))
    fasta, log = [], []
    gapcount = filled = 0
    for comment, seq in records:
        out = ""
        prev = 0
        i = 0
        n = len(seq)
        while i < n:
            if seq[i] in "Nn":
                gapcount += 1
                lmf = min(i + k - prev, max_fuz)
                ks = i - k - lmf
                gap = 0
                while i < n and seq[i] in "Nn":
                    i += 1
                    gap += 1
                rmf = min(n - (i + k), max_fuz)
                ok = (i + k + rmf <= n) and rmf >= 0
                if ok and any(c in "Nn" for c in seq[i:i + k + rmf]):
                    ok = False
                if ks >= prev and ok:
                    info = Info()
                    cnt, lf, rf, fill, sub = fill_gap(G, rng, seq[ks:ks + k + lmf], seq[i:i + k + rmf], gap, d_err,
                                                      lmf, rmf, skip_confident, all_paths, True, info)
                    log.append(info.log)
                    tail = fill_string(fill, lmf - lf) if fill is not None else ""
                    fstart = len(out) + ks + k + lmf - lf - prev
                    log.append(stats_line(comment, fstart, ks + k + lmf, i, cnt, tail, k, lmf, rmf, lf, rf,
                                          skip_confident, unique, sub, gap))
                    if cnt > 0 and (not unique or cnt == 1):
                        filled += 1
                        out = seq[prev:ks + k + lmf - lf] + tail
                        out = out[:len(out) - k]
                        i += rf
                    else:
                        out = out + seq[prev:ks + k + lmf + gap]
                else:
                    out = out + seq[prev:max(prev, ks + k + lmf + gap)]
                prev = i
            else:
                i += 1
        out = out + seq[prev:]
        fasta.append(">%s\n%s\n" % (comment, out))
    log.append("Filled %d gaps out of %d\n" % (filled, gapcount))
    return "".join(fasta), "".join(log), filled, gapcount


def brute_force_walks(G, kl, kr, g, lmf, rmf, j, length):
    """Number of walks of exactly `length` edges that start at any left seed
    (seed d starts with d steps already spent) and end at right k-mer j, WITHOUT
    pruning and with the reference's seed-overwrite rule (Q6) ignored: only valid
    as a check when seeds are not reachable from earlier seeds at equal depth
    other than along the flank itself."""
    k = G.k
    tgt = G.node(kr[j:])
    # DP by plain enumeration (exponential in branching; toy graphs only)
    level = {}
    x = G.node(kl)
    if G.contains(x):
        level[x] = 1
    for d in range(1, length + 1):
        nxt = {}
        for n, c in level.items():
            for v in G.succ(n):
                nxt[v] = nxt.get(v, 0) + c
        if d <= lmf:
            s = G.node(kl[d:])
            if G.contains(s):
                nxt[s] = 1
        level = nxt
    return level.get(tgt, 0)
