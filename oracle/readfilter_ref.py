"""oracle/readfilter_ref.py — Python restatement of the reference's ReadFilter
(/root/reference/src/ReadFilter.cpp:105-415), pass by pass in the reference's own order.
TEST INFRASTRUCTURE ONLY (the checker of gap2seq_amd/csrc/readfilter.cpp + bam.cpp); parity
unpinned by the reference, which ships no BAM fixtures and cannot be built here (GATB, htslib).
What is recalled rather than read from sources in the image: htslib's region iterator (negative
start -> 0, end < start -> no iterator, overlap = same tid, pos < end, bam_endpos > start),
GATB's Bloom (bit `hash % size` of a `size`-bit array) and libstdc++'s std::hash<std::string>
(64-bit Murmur-style _Hash_bytes, seed 0xc70f6907) — the last one is pinned against the real
std::hash by tests/test_readfilter.py through the product library."""
import struct
import zlib

M64 = (1 << 64) - 1

FUNMAP, FMUNMAP, FREVERSE, FREAD1 = 4, 8, 16, 64


def std_hash(b):
    """libstdc++ _Hash_bytes(ptr, len, 0xc70f6907) for 64-bit size_t."""
    mul = ((0xC6A4A793 << 32) + 0x5BD1E995) & M64
    n = len(b)
    h = (0xC70F6907 ^ (n * mul)) & M64
    al = n & ~7
    for i in range(0, al, 8):
        d = (struct.unpack_from("<Q", b, i)[0] * mul) & M64
        d ^= d >> 47
        d = (d * mul) & M64
        h ^= d
        h = (h * mul) & M64
    if n & 7:
        d = int.from_bytes(b[al:], "little")
        h ^= d
        h = (h * mul) & M64
    h ^= h >> 47
    h = (h * mul) & M64
    h ^= h >> 47
    return h


def bgzf_inflate(data):
    """every gzip member of a BGZF file, concatenated"""
    out = []
    while data:
        d = zlib.decompressobj(31)
        out.append(d.decompress(data))
        if not d.eof:
            raise ValueError("truncated BGZF member")
        data = d.unused_data
    return b"".join(out)


class Rec:
    __slots__ = ("tid", "pos", "flag", "name", "cigar", "seq4", "l_seq")

    def end_pos(self):  # htslib bam_endpos
        rlen = 0
        if not (self.flag & FUNMAP):
            for ln, op in self.cigar:
                if op in (0, 2, 3, 7, 8):
                    rlen += ln
        return self.pos + (rlen if rlen else 1)


def parse_bam(data):
    raw = bgzf_inflate(data)
    if raw[:4] != b"BAM\x01":
        raise ValueError("not a BAM file")
    l_text, = struct.unpack_from("<i", raw, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, o)
        refs.append(raw[o + 4:o + 4 + l_name].split(b"\0")[0].decode())
        o += 4 + l_name + 4
    recs = []
    while o < len(raw):
        bs, = struct.unpack_from("<i", raw, o)
        tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, _ntid, _npos, _tlen = struct.unpack_from("<iiBBHHHiiii", raw, o + 4)
        p = o + 36
        r = Rec()
        r.tid, r.pos, r.flag, r.l_seq = tid, pos, flag, l_seq
        r.name = raw[p:p + l_name].split(b"\0")[0]
        p += l_name
        r.cigar = [(w >> 4, w & 15) for w in struct.unpack_from("<%dI" % n_cig, raw, p)]
        p += 4 * n_cig
        r.seq4 = raw[p:p + (l_seq + 1) // 2]
        recs.append(r)
        o += 4 + bs
    return refs, recs


def _complement(n):  # ReadFilter.cpp:105-120
    return {1: 8, 2: 4, 4: 2, 8: 1}.get(n, 15)


def _to_string(r):  # ReadFilter.cpp:122-161 (case 0x15 never matches a 4-bit code; everything else is N)
    def code(i):
        return (r.seq4[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15
    out = []
    rev = bool(r.flag & FREVERSE)
    for i in range(r.l_seq):
        c = _complement(code(r.l_seq - 1 - i)) if rev else code(i)
        out.append({1: "A", 2: "C", 4: "G", 8: "T"}.get(c, "N"))
    return "".join(out)


def _name(r):  # :165-168
    return r.name + (b"/1" if r.flag & FREAD1 else b"/2")


def _mate(r):  # :170-173
    return r.name + (b"/2" if r.flag & FREAD1 else b"/1")


class Bloom:  # GATB Bloom with the reference's seed-ignoring hash1 (:28-33): one bit per item
    def __init__(self, size):
        self.size, self.bits = size, set()

    def insert(self, s):
        self.bits.add(std_hash(s) % self.size)

    def contains(self, s):
        return (std_hash(s) % self.size) in self.bits


def _query(recs, tid, beg, end, warn):  # sam_iterator(io, tid, start, end) (:184-191) + next() (:213-220)
    if beg < 0:
        beg = 0
    if tid < 0 or end < beg:
        warn.append("WARNING: SAM iterator is NULL!\n")
        return []
    if beg >= end:  # (reg2bins finds no bin for an empty region: the iterator exists and ends at once)
        return []
    return [r for r in recs if r.tid == tid and r.pos < end and r.end_pos() > beg]


def read_filter(data, mean, std_dev, scaffold, breakpoint, gap_length=-1, flank_length=-1, unmapped_only=False):
    """-> (fasta text, stdout text, stderr text)   ReadFilter::execute, :344-415"""
    refs, recs = parse_bam(data)
    warn, out = [], []
    num = len(recs)  # count_reads, :225-241
    read_length = max([r.l_seq for r in recs] + [0])
    bloom = Bloom(5 * num)  # :371
    extracted = 0

    def emit(r):  # print_fasta, :281-290
        out.append(">" + _name(r).decode() + "\n" + _to_string(r) + "\n")

    if not unmapped_only:
        tid = refs.index(scaffold) if scaffold in refs else -1  # :381
        left_start = breakpoint - (mean + 3 * std_dev + 2 * read_length)  # :384-385
        left_end = breakpoint - (mean - 3 * std_dev + read_length)
        for r in _query(recs, tid, left_start, left_end, warn):  # process_mates, :300-310
            if r.flag & FMUNMAP:
                bloom.insert(_name(r))
        right_start = breakpoint + (mean + 3 * std_dev + read_length) + gap_length  # :388-389, as written
        right_end = breakpoint + (mean - 3 * std_dev + read_length) + gap_length
        for r in _query(recs, tid, right_start, right_end, warn):
            if r.flag & FMUNMAP:
                bloom.insert(_name(r))
        for r in recs:  # find_mates, :313-323
            if bloom.contains(_mate(r)):
                emit(r)
                extracted += 1
        if flank_length != -1:  # :396-400, process_region :293-297
            for r in _query(recs, tid, breakpoint - flank_length, breakpoint + flank_length + gap_length, warn):
                if not bloom.contains(_name(r)):
                    emit(r)
                    extracted += 1
    if unmapped_only:  # :403-405, process_unmapped :326-337
        for r in recs:
            if (r.flag & FUNMAP) and not bloom.contains(_name(r)):
                emit(r)
                extracted += 1
    return "".join(out), "Extracted %d out of %d reads\n" % (extracted, num), "".join(warn)
