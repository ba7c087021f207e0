"""ReadFilter (/root/reference/src/ReadFilter.cpp:344-415): the product's BAM reader + filter
(gap2seq_amd/csrc/bam.cpp, readfilter.cpp, the `ReadFilter` binary) against the Python restatement in
oracle/readfilter_ref.py on simulated paired-read libraries.  CPU only: the filter is host work."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import bamwriter as BW  # noqa: E402
import readfilter_ref as REF  # noqa: E402
from gap2seq_amd import lib as P  # noqa: E402

BIN = os.path.join(ROOT, "gap2seq_amd", "ReadFilter")


def _both(bam, **kw):
    got = P.filter_reads(bam, **kw)
    want = REF.read_filter(bam, kw["mean"], kw["std_dev"], kw["scaffold"], kw["breakpoint"], kw.get("gap_length", -1),
                           kw.get("flank_length", -1), kw.get("unmapped_only", False))
    assert got[0] == want[0]
    assert got[1] == want[1]
    assert got[2] == want[2]
    return got


def test_golden_cases():
    """tests/golden/readfilter_cases.json (make_readfilter_golden.py): a committed BAM file and the committed
    answers, against the product AND against the restatement."""
    import base64
    import json
    g = json.load(open(os.path.join(HERE, "golden", "readfilter_cases.json")))
    bam = base64.b64decode(g["bam_base64"])
    assert len(g["calls"]) >= 6
    n = 0
    for c in g["calls"]:
        kw = c["args"]
        got = P.filter_reads(bam, **kw)
        assert got[:3] == (c["fasta"], c["stdout"], c["stderr"]) and got[4] == g["records"]
        want = REF.read_filter(bam, kw["mean"], kw["std_dev"], kw["scaffold"], kw["breakpoint"], kw.get("gap_length", -1),
                               kw.get("flank_length", -1), kw.get("unmapped_only", False))
        assert want == (c["fasta"], c["stdout"], c["stderr"])
        n += got[3]
    assert n > 10


def test_std_hash_restatement_is_libstdcxx(tmp_path):
    """The checker's std::hash<std::string> against the compiler's own."""
    names = ["", "a", "r00001/1", "r00001/2", "abcdefgh", "abcdefghi", "x" * 31, "read:with:colons/2", "0123456789abcdef"]
    src = tmp_path / "h.cpp"
    src.write_text('#include <cstdio>\n#include <functional>\n#include <string>\nint main(int c, char** v) {'
                   ' for (int i = 1; i < c; i++) printf("%zu\\n", std::hash<std::string>{}(std::string(v[i]))); }\n')
    exe = tmp_path / "h"
    subprocess.check_call(["g++", "-O1", "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)] + names[1:]).decode().split()
    assert [int(x) for x in out] == [REF.std_hash(n.encode()) for n in names[1:]]


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
@pytest.mark.parametrize("block", [65280, 700, 97])
def test_filter_matches_the_restatement(seed, block):
    refs, recs, _ = BW.simulate_library(seed, pairs=300 if block > 100 else 60)
    bam = BW.bam_bytes(refs, recs, block=block, extra_first=(seed % 2 == 0))
    total = len(recs)
    n_out = 0
    for scaffold in ("scaf0", "scaf1"):
        for kw in (dict(mean=300, std_dev=20, breakpoint=1400, gap_length=200, flank_length=100),
                   dict(mean=300, std_dev=20, breakpoint=1400, gap_length=200),
                   dict(mean=300, std_dev=0, breakpoint=1400, gap_length=200, flank_length=30),
                   dict(mean=250, std_dev=40, breakpoint=1400, gap_length=0, flank_length=0),
                   dict(mean=300, std_dev=20, breakpoint=100, gap_length=10, flank_length=200),  # windows left of 0
                   dict(mean=300, std_dev=20, breakpoint=1400, flank_length=50)):  # -gap-length default -1
            got = _both(bam, scaffold=scaffold, **kw)
            assert got[4] == total
            n_out += got[3]
    assert n_out > 0
    got = _both(bam, mean=300, std_dev=20, scaffold="0", breakpoint=0, gap_length=0, unmapped_only=True)
    assert got[3] > 0 and got[2] == ""


def test_mates_of_reads_an_insert_away_are_found():
    """What the filter is for: the unmapped ends whose mates align an insert size left of the gap come out,
    as sequenced, under their own end's name (and exactly once per read when names do not collide)."""
    refs, recs, facts = BW.simulate_library(11, n_scaffolds=1, pairs=500, unmapped_pairs=5, ambiguous=0.0)
    bam = BW.bam_bytes(refs, recs)
    fasta, log, warn, extracted, total = _both(bam, mean=300, std_dev=20, scaffold="scaf0", breakpoint=1400, gap_length=200)
    assert warn == "WARNING: SAM iterator is NULL!\n"  # the right-hand window (ReadFilter.cpp:388-389)
    by_name = {(n, w): (f, s) for n, w, f, s in facts}
    lines = fasta.splitlines()
    assert len(lines) == 2 * extracted and extracted > 0
    unmapped_hits = 0
    for h, s in zip(lines[0::2], lines[1::2]):
        name, which = h[1:].rsplit("/", 1)
        flag, orig = by_name[(name, int(which))]
        assert s == orig
        unmapped_hits += bool(flag & 4)
    assert unmapped_hits > 0
    assert log == "Extracted %d out of %d reads\n" % (extracted, total)


def test_reverse_strand_and_ambiguity_codes():
    refs = [("s", 1000)]
    recs = [BW.record("p", 1 | 64 | 8, 0, 100, "8M", "ACGTNMRA", 0, 100),           # forward, mate unmapped
            BW.record("p", 1 | 128 | 4, 0, 100, "", "AACCGGTTY", 0, 100),            # its unmapped mate, placed with it
            BW.record("q", 1 | 64 | 16 | 8, 0, 120, "8M", "AACGTNKC", 0, 120),      # reverse strand
            BW.record("q", 1 | 128 | 4 | 16, 0, 120, "", "ACGGT", 0, 120)]          # unmapped but flagged reverse
    bam = BW.bam_bytes(refs, recs)
    # window [bp - (mean + 2*rl), bp - (mean + rl)) with read length 9
    fasta, log, warn, extracted, total = _both(bam, mean=100, std_dev=0, scaffold="s", breakpoint=215, gap_length=50)
    assert total == 4
    fasta_all = _both(bam, mean=0, std_dev=0, scaffold="s", breakpoint=0, gap_length=0, unmapped_only=True)[0]
    assert fasta_all == ">p/2\nAACCGGTTN\n>q/2\nACCGT\n"
    fl = _both(bam, mean=100, std_dev=0, scaffold="s", breakpoint=110, gap_length=0, flank_length=20)
    assert ">q/1\nGNNACGTT\n" in fl[0] and ">p/1\nACGTNNNA\n" in fl[0]


def test_region_overlap_uses_the_alignment_end():
    refs = [("s", 5000)]
    recs = [BW.record("a", 64, 0, 1000, "50M", "A" * 50),
            BW.record("b", 64, 0, 1000, "10M100N40M", "C" * 50),      # spans to 1150
            BW.record("c", 64, 0, 1000, "20S30M", "G" * 50),          # ends at 1030
            BW.record("d", 64, 0, 1000, "25M50D25M", "T" * 50),       # ends at 1100
            BW.record("e", 64 | 4, 0, 1120, "", "ACGT"),              # unmapped: one base at 1120
            BW.record("f", 64, 0, 1130, "4M", "ACGT")]
    bam = BW.bam_bytes(refs, recs)
    for bp, flank, expect in ((1060, 10, "bd"), (1035, 5, "abd"), (1120, 1, "be"), (1140, 9, "bf"), (1029, 0, "abcd")):
        got = _both(bam, mean=10000, std_dev=0, scaffold="s", breakpoint=bp, gap_length=0 if flank else 1, flank_length=flank)
        names = "".join(ln[1] for ln in got[0].splitlines() if ln.startswith(">"))
        assert names == expect, (bp, flank, names)


def test_name_collisions_in_a_small_filter_are_reproduced():
    """5 bits per record and ONE effective hash (ReadFilter.cpp:28-33): on small files names collide, and the
    reference then extracts reads nobody asked for / drops flank reads.  Both sides must agree on those."""
    extra = 0
    for seed in range(20, 32):
        refs, recs, _ = BW.simulate_library(seed, n_scaffolds=1, scaffold_len=1500, gap=(700, 100), pairs=25, unmapped_pairs=2)
        bam = BW.bam_bytes(refs, recs)
        got = _both(bam, mean=300, std_dev=30, scaffold="scaf0", breakpoint=700, gap_length=100, flank_length=60)
        _, parsed = REF.parse_bam(bam)
        wanted = {REF._name(r) for r in parsed}
        assert len(wanted) == len(parsed)
        # exact set semantics would extract: mates of window reads + flank reads
        lo, hi = 700 - (300 + 90 + 100), 700 - (300 - 90 + 50)
        exact = {REF._name(r) for r in parsed if r.tid == 0 and (r.flag & 8) and r.pos < hi and r.end_pos() > max(lo, 0)}
        n_exact = sum(1 for r in parsed if REF._mate(r) in exact) + \
            sum(1 for r in parsed if r.tid == 0 and r.pos < 860 and r.end_pos() > 640 and REF._name(r) not in exact)
        extra += got[3] != n_exact
    assert extra > 0  # the collisions do happen at this size, so the comparison above covered them


def test_empty_and_unknown_scaffold_and_threads():
    refs = [("s", 100)]
    bam = BW.bam_bytes(refs, [])
    got = _both(bam, mean=100, std_dev=10, scaffold="s", breakpoint=50, gap_length=5, flank_length=10)
    assert got[0] == "" and got[1] == "Extracted 0 out of 0 reads\n"
    refs, recs, _ = BW.simulate_library(5, pairs=1500)
    bam = BW.bam_bytes(refs, recs, block=600)  # ~500 blocks: several inflating threads
    got = _both(bam, mean=300, std_dev=20, scaffold="nosuch", breakpoint=1400, gap_length=200, flank_length=100)
    assert got[3] == 0 and got[2].count("WARNING") == 3
    one = P.filter_reads(bam, mean=300, std_dev=20, scaffold="scaf1", breakpoint=1400, gap_length=200, flank_length=100, threads=1)
    many = P.filter_reads(bam, mean=300, std_dev=20, scaffold="scaf1", breakpoint=1400, gap_length=200, flank_length=100, threads=7)
    assert one == many and one[3] > 0


def test_records_spanning_refills(tmp_path):
    """The reader inflates a window of blocks at a time and carries a cut record over to the next window; with
    G2S_BAM_CHUNK=1500 (normally 32 MB) every few records are cut."""
    refs, recs, _ = BW.simulate_library(9, pairs=400)
    bam = BW.bam_bytes(refs, recs, block=400)
    path = tmp_path / "lib.bam"
    path.write_bytes(bam)
    out = tmp_path / "r.fa"
    want = REF.read_filter(bam, 300, 20, "scaf1", 1400, 200, 100)
    for chunk, cands in (("1500", None), ("401", None), ("70000", None), ("70000", "3"), ("1500", "0")):
        env = dict(os.environ, G2S_BAM_CHUNK=chunk)
        if cands is not None:  # more window candidates than the filter remembers in its first pass: it reads the file again
            env["G2S_FILTER_MAX_CANDS"] = cands
        r = subprocess.run([BIN, "-reads", str(out), "-scaffold", "scaf1", "-breakpoint", "1400", "-flank-length", "100",
                            "-gap-length", "200", "-bam", str(path), "-mean", "300", "-std-dev", "20"], capture_output=True,
                           text=True, env=env)
        assert r.returncode == 0 and r.stdout == want[1] and out.read_text() == want[0], (chunk, cands)
        out.unlink()


def test_broken_files_are_reported():
    refs, recs, _ = BW.simulate_library(6, pairs=50)
    bam = BW.bam_bytes(refs, recs, block=500)
    kw = dict(mean=300, std_dev=20, scaffold="scaf0", breakpoint=1400, gap_length=200)
    for bad in (bam[:len(bam) // 2], bam[:-40], b"not a bam", bam[:200] + bytes([bam[200] ^ 0x55]) + bam[201:],
                BW.bgzf(b"BAX\x01" + b"\0" * 20)):
        with pytest.raises(P.G2SError) as e:
            P.filter_reads(bad, **kw)
        assert e.value.code == P.G2S_ERR_IO
    raw = REF.bgzf_inflate(bam)
    with pytest.raises(P.G2SError):  # the last record cut short inside a well-formed BGZF stream
        P.filter_reads(BW.bgzf(raw[:-3]), **kw)
    P.filter_reads(BW.bgzf(raw, eof=False), **kw)  # a missing end-of-file marker is tolerated (htslib warns only)


def test_garbled_records_never_crash():
    """Bytes of the inflated stream changed at random and compressed again (so every block checksum is right and the
    record parser sees the damage): the filter either reports G2S_ERR_IO or returns text; whenever the checker
    can read the same bytes the two agree."""
    import random
    refs, recs, _ = BW.simulate_library(8, pairs=40, unmapped_pairs=3)
    raw = bytearray(REF.bgzf_inflate(BW.bam_bytes(refs, recs)))
    rng = random.Random(99)
    kw = dict(mean=300, std_dev=20, scaffold="scaf0", breakpoint=1400, gap_length=200, flank_length=100)
    errors = agreed = 0
    for it in range(300):
        b = bytearray(raw)
        for _ in range(rng.randrange(1, 4)):
            i = rng.randrange(len(b))
            b[i] = rng.randrange(256) if rng.random() < 0.7 else (b[i] ^ (1 << rng.randrange(8)))
        if rng.random() < 0.2:
            del b[rng.randrange(len(b)):]
        bam = BW.bgzf(bytes(b), block=rng.choice([65280, 333]))
        try:
            got = P.filter_reads(bam, **kw)
        except P.G2SError as e:
            assert e.code == P.G2S_ERR_IO
            errors += 1
            continue
        try:
            want = REF.read_filter(bam, 300, 20, "scaf0", 1400, 200, 100)
        except Exception:
            continue  # (the checker is less careful about broken layouts than the product)
        if all(32 <= ord(c) < 127 or c == "\n" for c in want[0]):
            assert got[:3] == want
            agreed += 1
    assert errors > 10 and agreed > 50


def test_binary_is_a_drop_in(tmp_path):
    """The wrapper's two calls (Gap2Seq.py:64-72 and :145-149): same argv, output file only when reads came out."""
    refs, recs, _ = BW.simulate_library(7, pairs=300)
    bam_path = tmp_path / "lib.bam"
    bam = BW.bam_bytes(refs, recs, block=4000)
    bam_path.write_bytes(bam)
    out = tmp_path / "tmp.reads.7.0"
    argv = [BIN, "-reads", str(out), "-scaffold", "scaf1", "-breakpoint", "1400", "-flank-length", "80", "-gap-length", "200",
            "-bam", str(bam_path), "-mean", "300", "-std-dev", "20"]
    r = subprocess.run(argv, capture_output=True, text=True)
    want = REF.read_filter(bam, 300, 20, "scaf1", 1400, 200, 80)
    assert r.returncode == 0 and r.stdout == want[1] and r.stderr == want[2]
    assert out.read_text() == want[0]
    r = subprocess.run(argv + ["-fasta-width", "20"], capture_output=True, text=True)
    seqs = [ln for ln in out.read_text().splitlines() if not ln.startswith(">")]
    assert max(len(s) for s in seqs) == 20
    assert "".join(seqs) == "".join(ln for ln in want[0].splitlines() if not ln.startswith(">"))
    un = tmp_path / "tmp.reads.0.unmapped"
    r = subprocess.run([BIN, "-unmapped-only", "-scaffold", "0", "-breakpoint", "0", "-gap-length", "0", "-reads", str(un),
                        "-bam", str(bam_path), "-mean", "300", "-std-dev", "20"], capture_output=True, text=True)
    want = REF.read_filter(bam, 300, 20, "0", 0, 0, -1, True)
    assert r.returncode == 0 and r.stdout == want[1] and un.read_text() == want[0]
    # nothing extracted: no file (Gap2Seq.py:151-153)
    none = tmp_path / "none.fa"
    r = subprocess.run([BIN, "-reads", str(none), "-scaffold", "nosuch", "-breakpoint", "5", "-bam", str(bam_path), "-mean", "300",
                        "-std-dev", "20"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("Extracted 0 out of") and not none.exists()
    # unreadable alignments: the reference's message, status 0 (ReadFilter.cpp:361-365)
    r = subprocess.run([BIN, "-reads", str(none), "-scaffold", "s", "-breakpoint", "5", "-bam", str(tmp_path / "missing.bam"),
                        "-mean", "300", "-std-dev", "20"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr.startswith("Error loading alignments") and not none.exists()
    r = subprocess.run([BIN, "-bogus"], capture_output=True, text=True)
    assert r.returncode != 0 and "Unknown parameter" in r.stdout
