"""tests/bamwriter.py — writes small BAM files for the read filter tests (SAM specification 4.1 BGZF, 4.2 BAM),
and simulates a paired-read library aligned to scaffolds with gaps.  There is no samtools/htslib in the image;
the files are read back by the product (gap2seq_amd/csrc/bam.cpp) and by the checker (oracle/readfilter_ref.py),
two independent parsers."""
import random
import struct
import zlib

SEQ_CODES = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"
BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bgzf_block(payload, level=6):
    assert len(payload) <= 65280
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = c.compress(payload) + c.flush()
    bsize = len(cdata) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + cdata +
            struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


def bgzf(raw, block=65280, eof=True, extra_first=False):
    """`raw` cut into members of `block` inflated bytes (small blocks make records span members)."""
    out = []
    for o in range(0, len(raw), block):
        out.append(bgzf_block(raw[o:o + block]))
    if not raw:
        out.append(bgzf_block(b""))
    if extra_first:  # another extra subfield in front of 'BC' in the first member (allowed by RFC 1952)
        b = out[0]
        xlen, = struct.unpack_from("<H", b, 10)
        bs, = struct.unpack_from("<H", b, 16)
        sub = b"XY\x03\x00abc"
        out[0] = b[:10] + struct.pack("<H", xlen + len(sub)) + sub + b"BC\x02\x00" + struct.pack("<H", bs + len(sub)) + b[18:]
    if eof:
        out.append(BGZF_EOF)
    return b"".join(out)


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def parse_cigar(text):
    ops, n = [], ""
    for ch in text:
        if ch.isdigit():
            n += ch
        else:
            ops.append((int(n), CIGAR_OPS.index(ch)))
            n = ""
    return ops


def record(name, flag, tid, pos, cigar, seq, mtid=-1, mpos=-1, tlen=0, mapq=30, qual=None):
    ops = parse_cigar(cigar) if isinstance(cigar, str) else cigar
    nm = name.encode() + b"\0"
    rlen = sum(n for n, op in ops if op in (0, 2, 3, 7, 8)) if not (flag & 4) else 0
    b = _reg2bin(max(pos, 0), max(pos, 0) + (rlen or 1))
    codes = [SEQ_CODES.index(c) if c in SEQ_CODES else 15 for c in seq.upper()]
    if len(codes) % 2:
        codes.append(0)
    packed = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    q = bytes(qual) if qual is not None else b"\xff" * len(seq)
    body = (struct.pack("<iiBBHHHiiii", tid, pos, len(nm), mapq, b, len(ops), flag, len(seq), mtid, mpos, tlen) + nm +
            b"".join(struct.pack("<I", (n << 4) | op) for n, op in ops) + packed + q)
    return struct.pack("<i", len(body)) + body


def bam_bytes(refs, records, text="@HD\tVN:1.6\tSO:coordinate\n", **bgzf_args):
    """refs: [(name, length)], records: bytes from record()"""
    hdr_text = text + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    raw = b"BAM\x01" + struct.pack("<i", len(hdr_text)) + hdr_text.encode() + struct.pack("<i", len(refs))
    for name, ln in refs:
        raw += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    raw += b"".join(records)
    return bgzf(raw, **bgzf_args)


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def simulate_library(seed, n_scaffolds=2, scaffold_len=3000, gap=(1400, 200), read_len=50, mean=300, sd=20, pairs=400,
                     unmapped_pairs=20, ambiguous=0.02):
    """Paired reads from random genomes whose middle [gap_at, gap_at+gap_len) is missing from the scaffold: pairs with
    one end inside the gap have that end unmapped (placed at its mate, as aligners do).  Returns (refs, records sorted
    by coordinate, facts) where facts lists for every read its name, end, flag and original sequence."""
    rng = random.Random(seed)
    gap_at, gap_len = gap
    refs = [("scaf%d" % i, scaffold_len) for i in range(n_scaffolds)]
    recs, facts = [], []
    uid = 0
    for tid in range(n_scaffolds):
        genome = "".join(rng.choice("ACGT") for _ in range(scaffold_len))
        for _ in range(pairs):
            ins = max(2 * read_len, int(rng.gauss(mean, sd)))
            a = rng.randrange(0, scaffold_len - ins)
            b = a + ins - read_len
            name = "r%05d" % uid
            uid += 1
            ends = []
            for which, start, rev in ((1, a, False), (2, b, True)):
                s = genome[start:start + read_len]
                if rng.random() < ambiguous:
                    i = rng.randrange(len(s))
                    s = s[:i] + rng.choice("NMRY") + s[i + 1:]
                inside = start + read_len > gap_at and start < gap_at + gap_len
                ends.append([which, start, rev, s, not inside])
            if rng.random() < 0.5:  # which end is read 1 varies
                ends[0][0], ends[1][0] = 2, 1
            m = [e for e in ends if e[4]]
            for e in ends:
                which, start, rev, s, mapped = e
                other = ends[1] if e is ends[0] else ends[0]
                flag = 1 | (64 if which == 1 else 128)
                if not mapped:
                    flag |= 4
                if not other[4]:
                    flag |= 8
                if mapped and rev:
                    flag |= 16
                if other[4] and other[2]:
                    flag |= 32
                if mapped:
                    pos, cig = start, "%dM" % len(s)
                    if rng.random() < 0.1:  # soft clips and deletions change the end position
                        cig = "5S%dM2D%dM" % (len(s) - 25, 20)
                    r_tid = tid
                elif m:  # unmapped, placed at the mapped mate
                    pos, cig, r_tid = m[0][1], "", tid
                else:
                    pos, cig, r_tid = -1, "", -1
                mt, mp = (tid, other[1]) if other[4] else ((tid, start) if mapped else (-1, -1))
                # the BAM stores the aligned strand: reverse alignments hold the reverse complement
                stored = s if not (flag & 16) else revcomp(s.replace("M", "N").replace("R", "N").replace("Y", "N"))
                recs.append(((r_tid if r_tid >= 0 else 1 << 30, pos), record(name, flag, r_tid, pos, cig, stored, mt, mp)))
                facts.append((name, which, flag, s))
    for _ in range(unmapped_pairs):
        name = "u%05d" % uid
        uid += 1
        for which in (1, 2):
            s = "".join(rng.choice("ACGT") for _ in range(read_len - (which == 2) * 7))
            flag = 1 | 4 | 8 | (64 if which == 1 else 128)
            recs.append(((1 << 30, -1), record(name, flag, -1, -1, "", s)))
            facts.append((name, which, flag, s))
    order = sorted(range(len(recs)), key=lambda i: (recs[i][0], i))
    return refs, [recs[i][1] for i in order], facts
