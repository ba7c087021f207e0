"""oracle/gapio_ref.py — Python restatement of the reference's GapCutter / GapMerger
(/root/reference/src/GapCutter.cpp:119-321, /root/reference/src/GapMerger.cpp:74-235).
TEST INFRASTRUCTURE ONLY (the checker of gap2seq_amd/csrc/gapio.cpp); parity unpinned by the
reference, which ships no fixtures for these tools and cannot be built here (GATB).  The variable
names d1, l1, d2, l2, d3 and the three cases are the reference's."""


def parse_fasta(text):
    recs = []
    for ln in text.splitlines():
        if ln.startswith(">"):
            recs.append([ln[1:], ""])
        elif recs and ln:
            recs[-1][1] += ln
    return [(c, s) for c, s in recs]


def _dist(seq, start):  # GapCutter.cpp:89-101
    n = 0
    while start + n < len(seq) and seq[start + n] not in "Nn":
        n += 1
    return n


def _gaplen(seq, start):  # GapCutter.cpp:103-115
    n = 0
    while start + n < len(seq) and seq[start + n] in "Nn":
        n += 1
    return n


def cut(text, k=31, fuz=10, mask=False, no_split=False):
    """-> (contigs [(comment, seq)], gaps [(comment, seq)], bed [(name, start, end)], counts (scaffolds, contigs, gaps))"""
    contigs, gaps, bed = [], [], []
    contig = gap = scaffold = 0
    for comment0, seq in parse_fasta(text):
        name = comment0.split(" ")[0]
        i = 0
        while i < len(seq):
            comment = "%s scaffold %d contig %d" % (comment0, scaffold, contig)
            gcomment = "%s gap %d" % (comment, gap)
            d1 = _dist(seq, i)
            l1 = _gaplen(seq, i + d1)
            if d1 > 0 and l1 == 0:  # :189
                contigs.append((comment, seq[i:]))
                contig += 1
                i = len(seq)
                continue
            if d1 < k:  # :198
                contigs.append((comment, seq[i:i + d1 + l1]))
                contig += 1
                i += d1 + l1
                continue
            flank1 = min(d1, k + fuz)
            d2 = _dist(seq, i + d1 + l1)
            l2 = _gaplen(seq, i + d1 + l1 + d2)
            d3 = _dist(seq, i + d1 + l1 + d2 + l2)
            if d2 >= 2 * k or (d2 >= k and d3 == 0):  # case 1, :212
                flank2 = min(d2, k + fuz) if d2 >= 2 * k else d2
                gaps.append((gcomment, seq[i + d1 - flank1:i + d1 + l1 + flank2]))
                contigs.append((gcomment, seq[i:i + d1 - flank1]))
                bed.append((name, i + d1 - flank1, i + d1 + l1 + flank2))
                gap += 1
                contig += 1
                i += d1 + l1 + flank2
                continue
            if d2 >= k:  # case 2, :236
                if not no_split and d3 >= k:
                    flank3 = min(d3, k + fuz)
                    gaps.append((gcomment + " split 1", seq[i + d1 - flank1:i + d1 + l1 + d2]))
                    gaps.append((gcomment + " split 2 %d" % k, seq[i + d1 + l1 + d2 - k:i + d1 + l1 + d2 + l2 + flank3]))
                    contigs.append((gcomment, seq[i:i + d1 - flank1]))
                    bed.append((name, i + d1 - flank1, i + d1 + l1 + d2))
                    bed.append((name, i + d1 + l1, i + d1 + l1 + d2 + l2 + flank3))
                    gap += 1
                    contig += 1
                    i += d1 + l1 + d2 + l2 + flank3
                else:
                    flank2 = min(d2, k + fuz)
                    gaps.append((gcomment, seq[i + d1 - flank1:i + d1 + l1 + flank2]))
                    contigs.append((gcomment, seq[i:i + d1 - flank1]))
                    bed.append((name, i + d1 - flank1, i + d1 + l1 + flank2))
                    gap += 1
                    contig += 1
                    i += d1 + l1 + flank2
                continue
            lsum, dn = l1 + d2 + l2, d3  # case 3, :281
            while 0 < dn < k:
                lsum += dn + _gaplen(seq, i + d1 + lsum + dn)
                dn = _dist(seq, i + d1 + lsum)
            if dn < k:
                contigs.append((comment, seq[i:]))
                contig += 1
                i = len(seq)
                continue
            if mask:
                flank2 = min(dn, k + fuz)
                gaps.append((gcomment, seq[i + d1 - flank1:i + d1] + "n" * lsum + seq[i + d1 + lsum:i + d1 + lsum + flank2]))
                contigs.append((gcomment, seq[i:i + d1 - flank1]))
                bed.append((name, i + d1 - flank1, i + d1 + lsum + flank2))
                gap += 1
                contig += 1
                i += d1 + lsum + flank2
            else:
                contigs.append((comment, seq[i:i + d1 + lsum]))
                contig += 1
                i += d1 + lsum
        scaffold += 1
    return contigs, gaps, bed, (scaffold, contig, gap)


def _index(comment, marker, until):
    at = comment.find(marker)
    if at < 0:
        return -1
    frm = at + len(marker)
    to = comment.find(until) if until else -1
    return int(comment[frm:to] if to >= 0 else comment[frm:])


def merge(contigs, gaps):
    """contigs, gaps: [(comment, seq)] -> (scaffolds [(comment, seq)], counts (contigs, gaps, scaffolds))"""
    out = []
    scaffold, scomment, sidx = "", contigs[0][0] if contigs else "", 0
    nc = ng = 0

    def emit():
        m = scomment.find(" scaffold ")
        out.append((scomment if m < 0 else scomment[:m], scaffold))

    for comment, seq in contigs:
        cs = _index(comment, " scaffold ", " contig ")
        gi = _index(comment, " gap ", " split ")
        nc += 1
        if cs != sidx:
            emit()
            scaffold, scomment, sidx = "", comment, cs
        scaffold += seq
        if gi != -1:
            first = second = ""
            for gcomment, gseq in gaps:
                if _index(gcomment, " gap ", " split ") != gi:
                    continue
                sp = gcomment.find(" split ")
                if sp < 0:
                    first = gseq
                    break
                if int(gcomment[sp + 7:sp + 8]) == 1:
                    first = gseq
                else:
                    second = gseq[int(gcomment[sp + 9:]):]
                if first and second:
                    break
            scaffold += first + second
            ng += 1
    if scaffold:
        emit()
        sidx += 1
    return out, (nc, ng, sidx)
